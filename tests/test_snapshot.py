"""JSON snapshots in the reference's wire format: reader and writer of agarcl_amd/snapshot.py against
files written / states produced by the real reference (tests/golden/make_snapshot_golden.py), here on the
wave-emulation build of the kernel source; tests/test_gpu_parity.py repeats the replay on the HIP engine."""
import json
import os

import numpy as np
import pytest

from snapshot_cases import replay_snapshot_case, snapshot_cases, strip


@pytest.mark.parametrize("base", snapshot_cases(), ids=lambda p: os.path.basename(p))
def test_snapshot_golden_on_emulated_kernels(emu_lib, base):
    from agarcl_amd import _capi
    ok, msg = replay_snapshot_case(lambda n, **cfg: _capi.BatchedEngine(n, lib=emu_lib, **cfg), base)
    assert ok, msg


def test_map_order_emulation_matches_libstdcxx(oracle_lib):
    """map_order_after_inserts (product side) against the oracle's restatement, which test_stl_emulation.py pins to the
    real libstdc++: insertion of 0..n-1 into a cleared table for every bucket state a players map can be in."""
    from agarcl_amd import snapshot
    import ctypes as C
    L = oracle_lib.lib()
    for buckets, resize in [(1, 0), (13, 13), (29, 29), (59, 59)]:
        for n in range(1, 17):
            order, hb, hr = snapshot.map_order_after_inserts(list(range(n)), buckets, resize)
            keys = np.arange(n, dtype=np.int32); out = np.zeros(n, dtype=np.int32)
            bc = np.array([buckets], dtype=np.int32); nr = np.array([resize], dtype=np.int32)
            L.ora_hash_order(keys.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), bc.ctypes.data_as(C.c_void_p), nr.ctypes.data_as(C.c_void_p))
            assert [int(keys[i]) for i in order] == [int(x) for x in out], (buckets, n)
            assert (hb, hr) == (int(bc[0]), int(nr[0])), (buckets, n)


def test_reference_loads_our_file_and_writer_matches_reference(emu_lib, ref_lib, tmp_path):
    """Both directions live against the reference build: (1) same state -> same JSON (before any load, so colours of
    bots and the mode_number quirk are covered too); (2) the reference loads a file written by this package and both
    continue in lock-step."""
    from agarcl_amd import _capi, snapshot
    from oracle import blob
    cfg = dict(num_agents=2, arena_size=220, num_pellets=90, num_viruses=4, num_bots=3, mode=0)
    na = 2
    ref = ref_lib.RefEnv(**cfg); ref.seed(21); ref.reset(True)
    eng = _capi.BatchedEngine(1, lib=emu_lib, **cfg); eng.seed(np.array([21], np.uint32)); eng.reset(reset_ids=True)
    rng = np.random.RandomState(4)
    for t in range(90):
        dxdy = rng.uniform(-1, 1, (na, 2)).astype(np.float32); act = rng.randint(0, 3, na).astype(np.int32)
        ref.take_actions(dxdy, act); ref.step(); eng.set_actions(dxdy[None], act[None]); eng.step()
    assert blob.diff(ref.dump(), eng.dump(0)) is None
    p_ref, p_our = str(tmp_path / "ref.json"), str(tmp_path / "our.json")
    ref.save_json(p_ref)
    wcfg = dict(num_agents=na, ticks_per_step=4, arena_size=220, num_bots=3, reward_type=True, c_death=0, mode_number=0, pellet_regen=True)
    ours = snapshot.save_arena(eng, 0, wcfg)
    open(p_our, "w").write(snapshot.dumps(ours))
    a, b = json.load(open(p_ref)), json.load(open(p_our))
    for pa, pb in zip(a["players"], b["players"]):      # agents' colours are random draws the engine does not keep
        if not pa["is_bot"]:
            for ca, cb in zip(pa["cells"], pb["cells"]):
                cb["color"] = ca["color"]
    assert a == b
    # (2) the reference reads OUR file; a second engine arena reads it too; both continue identically
    ref2 = ref_lib.RefEnv(**cfg); ref2.seed(3); ref2.reset(True); ref2.load_json(p_our, reset_ids=True)
    eng2 = _capi.BatchedEngine(1, lib=emu_lib, **cfg); eng2.seed(np.array([3], np.uint32)); eng2.reset(reset_ids=True)
    snapshot.load_arena(eng2, 0, json.load(open(p_our)), reset_ids=True)
    assert blob.diff(ref2.dump(), eng2.dump(0)) is None
    for t in range(80):
        dxdy = rng.uniform(-1, 1, (na, 2)).astype(np.float32); act = rng.randint(0, 3, na).astype(np.int32)
        ref2.take_actions(dxdy, act); r = ref2.step(); eng2.set_actions(dxdy[None], act[None]); eng2.step()
        assert np.array_equal(np.array(r), eng2.rewards()[0]), t
    assert blob.diff(ref2.dump(), eng2.dump(0)) is None


def test_agarcl_module_mirror_save_load(emu_lib, tmp_path, monkeypatch):
    """GridEnvironment.save_env_state / load_env_state with the reference's semantics (reset() is a no-op after a
    load, BaseEnvironment.hpp:180-181; file errors are RuntimeErrors)."""
    from agarcl_amd import agarcl
    monkeypatch.setattr(agarcl, "_LIB", emu_lib)
    env = agarcl.GridEnvironment(1, 4, 200, True, 60, 4, 0, 1, 0, 6)
    env.seed(7); env.reset()
    for t in range(30):
        env.take_actions([(0.5, -0.25, t % 3)]); env.step()
    p = str(tmp_path / "s.json")
    env.save_env_state(p)
    snap = json.load(open(p))
    assert snap["mode_number"] == 0 and snap["seed"] == 7 and snap["players"][0]["name"] == "agent0" and len(snap["pellets"]) == snap["pellet_count"]
    env2 = agarcl.GridEnvironment(1, 4, 200, True, 60, 4, 0, 1, 0, 6)
    env2.seed(1); env2.reset(); env2.load_env_state(p)
    m_before = env2._engine.masses().copy()
    env2.reset()                                   # no-op after a load
    assert np.array_equal(env2._engine.masses(), m_before)
    env2.take_actions([(0.1, 0.1, 0)]); r = env2.step()
    assert len(r) == 1 and not env2.dones()[0]
    env2.save_env_state(p)
    assert strip(json.load(open(p)))["players"][0]["name"] == "agent0"
    with pytest.raises(RuntimeError):
        env2.load_env_state(str(tmp_path / "missing.json"))
    env.close(); env2.close()
