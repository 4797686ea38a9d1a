"""Round 6's default capacities (include/agarcl_batch.h cap_viruses / cap_foods): an arena created without viruses holds 16 virus slots (it never grows one;
only a loaded state can bring them), and a state that does not fit is refused loudly (AGARCL_E_CAPACITY), never truncated.  CPU: the emulation build."""
import numpy as np
import pytest


def _played(lib, steps, **cfg):
    from agarcl_amd import _capi
    eng = _capi.BatchedEngine(1, lib=lib, **cfg)
    eng.seed(np.array([5], np.uint32)); eng.reset(reset_ids=True)
    rng = np.random.RandomState(1)
    for t in range(steps):
        eng.set_actions(rng.uniform(-1, 1, (1, 1, 2)).astype(np.float32), rng.randint(0, 3, (1, 1)).astype(np.int32)); eng.step()
    return eng


def test_virus_free_arena_takes_a_state_with_few_viruses_and_refuses_one_with_many(emu_lib):
    from agarcl_amd import _capi
    base = dict(arena_size=400, num_pellets=200, mode=0)
    few = _played(emu_lib, 5, num_viruses=12, **base)
    many = _played(emu_lib, 5, num_viruses=30, **base)
    assert few.counts()[0][1] == 12 and many.counts()[0][1] == 30
    dst = _capi.BatchedEngine(1, lib=emu_lib, num_viruses=0, **base)
    dst.seed(np.array([9], np.uint32)); dst.reset(reset_ids=True)
    dst.load(few.dump(0), 0)                       # 12 <= 16 slots
    rng = np.random.RandomState(3)
    for t in range(20):                            # ... and both continue identically (the capacity is not part of the state)
        dxdy = rng.uniform(-1, 1, (1, 1, 2)).astype(np.float32); act = rng.randint(0, 3, (1, 1)).astype(np.int32)
        few.set_actions(dxdy, act); few.step(); dst.set_actions(dxdy, act); dst.step()
    assert np.array_equal(few.dump(0), dst.dump(0))
    assert dst.counts()[0][1] == few.counts()[0][1] >= 10      # (counts() is what the last step left)
    before = dst.dump(0).copy()
    with pytest.raises(_capi.AgarclError) as ei:   # 30 > 16: refused, the arena keeps what it had
        dst.load(many.dump(0), 0)
    assert "capacit" in str(ei.value).lower()
    assert np.array_equal(dst.dump(0), before)
    big = _capi.BatchedEngine(1, lib=emu_lib, num_viruses=0, cap_viruses=40, **base)   # an explicit capacity takes it
    big.seed(np.array([9], np.uint32)); big.reset(reset_ids=True)
    big.load(many.dump(0), 0)
    assert np.array_equal(big.dump(0), many.dump(0))
    for e in (few, many, dst, big): e.close()
