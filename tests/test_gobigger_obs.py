"""GoBigger observation: the padded tensors of agarcl_gobigger_obs and the object view derived from them
against the host restatement oracle/gobigger_oracle.py (parity UNPINNED: see that file), in lock-step over a rollout with
several players, splits, ejected foods and deaths."""
import numpy as np
import pytest

CFG = dict(num_agents=2, arena_size=220, num_pellets=400, num_viruses=6, num_bots=2, mode=0)


def _objects_equal(a, b):
    sa, sb = a.get_all_player_states(), b.get_all_player_states()
    assert set(sa) == set(sb)
    for pid in sa:
        x, y = sa[pid], sb[pid]
        assert x.get_score() == y.get_score() and x.get_team_name() == y.get_team_name(), pid
        for getter in ("get_food_infos", "get_virus_infos", "get_spore_infos", "get_clone_infos"):
            la, lb = getattr(x, getter)(), getattr(y, getter)()
            assert len(la) == len(lb), (pid, getter, len(la), len(lb))
            for u, v in zip(la, lb):
                assert np.float32(u.position.x) == np.float32(v.position.x) and np.float32(u.position.y) == np.float32(v.position.y)
                assert np.float32(u.radius) == np.float32(v.radius) and u.score == v.score
                if getter == "get_clone_infos":
                    assert tuple(np.float32(u.velocity)) == tuple(np.float32(v.velocity)) and u.owner == v.owner and u.teamId == v.teamId
                    assert (np.float32(u.direction) == np.float32(v.direction)) or (np.isnan(u.direction) and np.isnan(v.direction))


def _run(engine_cls, lib, steps):
    from agarcl_amd import gobigger
    from oracle import gobigger_oracle
    A = 3
    eng = engine_cls(A, lib=lib, **CFG) if lib is not None else engine_cls(A, **CFG)
    eng.seed(None, 50); eng.reset(reset_ids=True)
    mine = [gobigger.PlayerStates() for _ in range(A)]
    want = [gobigger.PlayerStates() for _ in range(A)]
    rng = np.random.RandomState(9)
    listed = 0
    for t in range(steps):
        eng.set_actions(rng.uniform(-1, 1, size=(A, 2, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 2)).astype(np.int32)); eng.step()
        if t % 7 and t < steps - 1:
            continue
        ten = eng.gobigger_obs(64, cap_food=448, cap_virus=16, cap_spore=64, cap_clone=32)
        assert ten["hdr"].shape == (A, 4, 8) and ten["food"].shape == (A, 4, 448, 4) and ten["clone"].shape == (A, 4, 32, 7)
        for a in range(A):
            gobigger.add_frame(mine[a], ten, a)
            gobigger_oracle.add_frame(want[a], eng.dump(a), 64)
            _objects_equal(mine[a], want[a])
            hdr = ten["hdr"][a]
            for k in range(hdr.shape[0]):                       # padding is zero, counts within capacity here
                nv, nf, ns, nc = hdr[k, 2:6]
                assert not ten["food"][a, k, nf:].any() and not ten["virus"][a, k, nv:].any() and not ten["spore"][a, k, ns:].any() and not ten["clone"][a, k, nc:].any()
                listed += nf + nv + ns + nc
    assert listed > 100
    # capacities smaller than the lists: counts stay true, rows are truncated in order
    small = eng.gobigger_obs(64, cap_food=4, cap_virus=1, cap_spore=1, cap_clone=2)
    big = eng.gobigger_obs(64, cap_food=448, cap_virus=16, cap_spore=64, cap_clone=32)
    assert np.array_equal(small["hdr"], big["hdr"]) and np.array_equal(small["food"], big["food"][:, :, :4]) and np.array_equal(small["clone"], big["clone"][:, :, :2])
    eng.close()


def test_gobigger_tensors_on_emulated_kernels(emu_lib):
    from agarcl_amd import _capi
    _run(_capi.BatchedEngine, emu_lib, 120)


@pytest.mark.gpu
def test_gobigger_tensors_hip(hip_engine_cls):
    _run(hip_engine_cls, None, 300)


@pytest.mark.gpu
def test_gobigger_tensors_device_resident(hip_engine_cls):
    """on_device: the five tensors are written straight into caller-owned HBM (torch tensors), no host copy"""
    import torch
    A = 64
    eng = hip_engine_cls(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
    eng.seed(None, 4); eng.reset(reset_ids=True)
    rng = np.random.RandomState(1)
    for t in range(30):
        eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32)); eng.step()
    dev = torch.device("cuda")
    hdr = torch.empty((A, 1, 8), dtype=torch.int32, device=dev)
    food, virus, spore = (torch.empty((A, 1, k, 4), dtype=torch.float32, device=dev) for k in (256, 64, 64))
    clone = torch.empty((A, 1, 32, 7), dtype=torch.float32, device=dev)
    eng.sync()
    eng.gobigger_obs(128, out_ptrs=(hdr.data_ptr(), food.data_ptr(), virus.data_ptr(), spore.data_ptr(), clone.data_ptr()))
    eng.sync()
    host = eng.gobigger_obs(128)
    assert np.array_equal(hdr.cpu().numpy(), host["hdr"]) and np.array_equal(food.cpu().numpy(), host["food"]) and np.array_equal(clone.cpu().numpy(), host["clone"], equal_nan=True)
    eng.close()
