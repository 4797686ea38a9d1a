"""The "ram" observation (BASELINE configs[0]): HIP kernel vs the host restatement oracle/ram_oracle.py -- bit-exact fp32.
Parity unpinned by construction: the reference has no ram observation (AgarioEnv.py:211 rejects it), the layout is the library's own
(include/agarcl_batch.h agarcl_ram_obs)."""
import numpy as np
import pytest

from lockstep import policy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60),   # BASELINE configs[0]'s population
    dict(num_agents=1, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6),
    dict(num_agents=3, arena_size=250, num_pellets=300, num_viruses=5, mode=6),
    dict(num_agents=1, arena_size=1000, num_pellets=1300, num_viruses=40, mode=0),   # more than 1024 pellets: the selection's general form
], ids=["c1_bots", "c3m6", "multi3", "many_pellets"])
def test_ram_obs_matches_host_restatement(hip_engine_cls, cfg):
    from oracle import ram_oracle
    A = 6
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 321); eng.reset(reset_ids=True)
    n = cfg["num_agents"]
    ks = dict(k_cells=16, k_pellets=16, k_viruses=8, k_others=16)
    for t in range(90):
        dx = np.zeros((A, n, 2), np.float32); ac = np.zeros((A, n), np.int32)
        for a in range(A):
            dx[a], ac[a] = policy(17 + a, t, n, True, 4)
        eng.set_actions(dx, ac); eng.step()
        if t % 15 == 14:
            got = eng.ram_obs(**ks)
            assert got.shape == (A, n, 4 + 48 + 32 + 24 + 48)
            for a in range(A):
                _, pl = eng.arena_words(a)
                want = ram_oracle.ram_obs(eng.dump(a), [int(pl[s, 15]) for s in range(eng.players)], **ks)     # PL_PID per slot
                g, w = got[a].copy(), want.copy()
                assert np.array_equal(np.isnan(g), np.isnan(w))          # (a dead agent's centre is 0 / 0: any NaN equals any NaN,
                g[np.isnan(g)] = 0; w[np.isnan(w)] = 0                   #  the payload bits of a NaN differ between x86 and the GPU)
                assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), (t, a, np.flatnonzero(g.view(np.uint32) != w.view(np.uint32))[:8])
    eng.close()


def test_ram_obs_device_buffer_and_sizes(hip_engine_cls):
    import torch
    eng = hip_engine_cls(4096, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
    eng.seed(None, 5); eng.reset(reset_ids=True)
    D = 4 + 3 * 4 + 2 * 8 + 3 * 2 + 3 * 4
    buf = torch.full((4096, 1, D), 7.0, dtype=torch.float32, device="cuda")
    assert eng.ram_obs(4, 8, 2, 4, out_ptr=buf.data_ptr()) == D
    eng.sync()
    host = eng.ram_obs(4, 8, 2, 4)
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), host.view(np.uint32))
    assert (host[:, 0, 3] == 1).all() and (host[:, 0, 2] == 25).all()          # one cell of mass 25 right after the reset
    d2 = host[:, 0, 16:32:2] ** 2 + host[:, 0, 17:32:2] ** 2                    # the 8 pellets come nearest first
    assert (np.diff(d2, axis=1) >= 0).all()
    from agarcl_amd._capi import AgarclError
    with pytest.raises(AgarclError):
        eng.ram_obs(k_cells=99)
    eng.close()
