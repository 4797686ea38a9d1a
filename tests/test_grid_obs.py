"""Grid observation (GridObservation::add_frame, GridEnvironment.hpp:91-123): engine vs the oracle's restatement,
int-exact.  NOTE: the oracle side of O1 is *unpinned* against the real reference (GridEnvironment.hpp is not
buildable here without OpenGL stand-ins)."""
import numpy as np
import pytest

from lockstep import policy

CFGS = [
    dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6),
    dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6),
    dict(arena_size=60, num_pellets=200, num_viruses=0, mode=0),   # tiny arena: large out-of-bounds band
    dict(arena_size=1000, num_pellets=1300, num_viruses=300, mode=6),   # more pellets / viruses than the kernel's up-front loads cover (1024 / 256)
]


def _run(engine_cls, oracle_lib, lib, cfg, steps=60, A=3):
    kw = dict(lib=lib) if lib is not None else {}
    eng = engine_cls(A, **kw, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    seeds = np.arange(31, 31 + A).astype(np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for o, s in zip(oras, seeds):
        o.seed(int(s)); o.reset(True)
    for t in range(steps):
        dxdy = np.zeros((A, 1, 2), np.float32); act = np.zeros((A, 1), np.int32)
        for a in range(A):
            dd, aa = policy(5 + a, t, 1, True, 8); dxdy[a, 0] = dd[0]; act[a, 0] = aa[0]
        eng.set_actions(dxdy, act); eng.step()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); oras[a].step()
        if t % 20 == 19:
            for G, flags in ((128, (True, True, True, True)), (32, (True, False, True, True)), (64, (False, True, False, True))):
                got = eng.grid_obs(G, *flags)
                for a in range(A):
                    want = oras[a].grid_obs(0, G, *flags)
                    assert got.shape[2:] == want.shape
                    assert np.array_equal(got[a, 0], want), (cfg, t, a, G, flags)
                assert (got[:, 0, 0] == -1).any() or cfg["arena_size"] > 300
    eng.close()


@pytest.mark.parametrize("cfg", CFGS)
def test_grid_obs_kernel_source_emulation(emu_lib, oracle_lib, cfg):
    from agarcl_amd import _capi
    _run(_capi.BatchedEngine, oracle_lib, emu_lib, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CFGS)
def test_grid_obs_hip(hip_engine_cls, oracle_lib, cfg):
    _run(hip_engine_cls, oracle_lib, None, cfg)


@pytest.mark.gpu
def test_grid_obs_device_buffer_4096(hip_engine_cls):
    """BASELINE config 5 shape: int32 [4096][1][8][128][128] written straight into a torch HBM tensor."""
    import torch
    A = 4096
    eng = hip_engine_cls(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    out = torch.empty((A, 1, 8, 128, 128), dtype=torch.int32, device="cuda")
    c = eng.grid_obs(128, out_ptr=out.data_ptr())
    eng.sync()
    assert c == 8
    host = eng.grid_obs(128)
    assert np.array_equal(out[:8].cpu().numpy(), host[:8])
    assert int(out[:, 0, 5].sum().item()) == 25 * A   # own-cell channel: one cell of mass 25 per agent
    eng.close()


@pytest.mark.gpu
def test_grid_obs_4096_sampled_arenas_vs_oracle(hip_engine_cls, oracle_lib):
    """Config 5 at full size, on the full rule set: 4096 mode-6 arenas stepped 40 times with random moves / feeds / splits; the persistent
    device tensor (undo-list path, rewritten every step) of sampled arenas -- first, last, tile edges -- equals the oracle's frame of the
    same arena advanced alone with the same seed and actions, every 10th step, word for word."""
    import torch
    A, G, steps = 4096, 128, 40
    cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
    sample = [0, 1, 63, 64, 1000, 2047, 2048, 4094, 4095]
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    oras = {a: oracle_lib.OraEnv(**cfg) for a in sample}
    for a, o in oras.items():
        o.seed(10000 + a); o.reset(True)
    out = torch.zeros((A, 1, 8, G, G), dtype=torch.int32, device="cuda")
    rng = np.random.RandomState(77)
    for t in range(steps):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = rng.randint(0, 3, size=(A, 1)).astype(np.int32)
        eng.set_actions(dxdy, act); eng.step()
        assert eng.grid_obs(G, out_ptr=out.data_ptr(), persistent=True) == 8
        for a, o in oras.items():
            o.take_actions(dxdy[a], act[a]); o.step()
        if t % 10 == 9:
            eng.sync()
            for a, o in oras.items():
                assert np.array_equal(out[a, 0].cpu().numpy(), o.grid_obs(0, G, True, True, True, True)), (t, a)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6),
                                 dict(num_agents=2, arena_size=250, num_pellets=500, num_viruses=10, num_bots=3, mode=0)])
def test_grid_obs_persistent_buffer(hip_engine_cls, cfg):
    """on_device = 2 (a persistent observation tensor: only what the previous call scattered is cleared) gives the tensor that a
    full clear gives, step after step -- agents that split, move and shrink their view; also after a switch of configuration and
    after another buffer was used in between."""
    import torch
    A, na = 48, cfg.get("num_agents", 1)
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 321); eng.reset(reset_ids=True)
    rng = np.random.RandomState(2)
    torch.cuda.synchronize()
    keep = torch.full((A, na, 8, 128, 128), 7, dtype=torch.int32, device="cuda")    # garbage: the first call must clear everything
    other = torch.full((A, na, 8, 128, 128), 9, dtype=torch.int32, device="cuda")
    fresh = torch.empty((A, na, 8, 128, 128), dtype=torch.int32, device="cuda")
    small = torch.full((A, na, 6, 64, 64), 3, dtype=torch.int32, device="cuda")   # 1 + cells + 2 viruses + 2 pellets
    torch.cuda.synchronize()
    for t in range(60):
        eng.set_actions(rng.uniform(-1, 1, size=(A, na, 2)).astype(np.float32), rng.randint(0, 3, size=(A, na)).astype(np.int32))
        eng.step()
        eng.grid_obs(128, out_ptr=keep.data_ptr(), persistent=True)
        eng.grid_obs(128, out_ptr=fresh.data_ptr())
        eng.sync()    # (the engine has its own stream)
        assert torch.equal(keep, fresh), t
        if t == 20:    # another persistent buffer in between: the engine must notice and clear `keep` fully next time
            eng.grid_obs(128, out_ptr=other.data_ptr(), persistent=True); eng.sync()
            assert torch.equal(other, fresh)
        if t == 40:    # another configuration into another buffer
            eng.grid_obs(64, True, False, True, True, out_ptr=small.data_ptr(), persistent=True)
            ref = torch.empty_like(small); eng.grid_obs(64, True, False, True, True, out_ptr=ref.data_ptr()); eng.sync()
            assert torch.equal(small, ref)
    eng.close()


@pytest.mark.gpu
def test_grid_obs_persistent_buffer_walls(hip_engine_cls):
    """The out-of-bounds channel of a persistent tensor is updated row- and column-wise from the previous call's signature:
    a 150 x 150 arena, where every view window (100 .. 300 wide) reaches over the walls and the mask moves with the agent every step;
    agents driven into corners and along walls, a masked reset that teleports some of them, and a stretch without any movement (no row or
    column changes: channel 0 is not touched).  Equal to the full-clear path after every step."""
    import torch
    A = 64
    eng = hip_engine_cls(A, arena_size=150, num_pellets=150, num_viruses=2, mode=0)
    eng.seed(None, 99); eng.reset(reset_ids=True)
    rng = np.random.RandomState(5)
    keep = torch.full((A, 1, 8, 128, 128), 7, dtype=torch.int32, device="cuda")
    fresh = torch.empty_like(keep)
    torch.cuda.synchronize()
    drive = np.zeros((A, 1, 2), np.float32)
    changed_steps = 0
    prev0 = None
    for t in range(90):
        if t % 15 == 0:      # a new heading per arena: corners, walls, diagonals
            drive = rng.choice([-1.0, 0.0, 1.0], size=(A, 1, 2)).astype(np.float32)
        if 45 <= t < 55:
            drive = np.zeros((A, 1, 2), np.float32)       # nobody moves: the mask stands
        eng.set_actions(drive, np.zeros((A, 1), np.int32)); eng.step()
        if t == 30:
            mask = (np.arange(A) % 3 == 0).astype(np.uint8); eng.reset(mask)      # some agents jump elsewhere
        eng.grid_obs(128, out_ptr=keep.data_ptr(), persistent=True)
        eng.grid_obs(128, out_ptr=fresh.data_ptr())
        eng.sync()
        assert torch.equal(keep, fresh), t
        ch0 = keep[:, 0, 0].clone()
        assert (ch0 == -1).any().item() and (ch0 == 0).any().item()               # the window really crosses the walls
        if prev0 is not None and not torch.equal(ch0, prev0):
            changed_steps += 1
        prev0 = ch0
    assert changed_steps > 40          # the incremental path had rows / columns to move in most steps
    eng.close()


@pytest.mark.gpu
def test_grid_obs_host_path_repeated(hip_engine_cls, oracle_lib):
    """The host-copy path goes through the engine's staging buffer, which it treats as persistent (incremental clear) -- unless
    something else used the buffer in between (screen / GoBigger host copies share it): same config every step, against the oracle."""
    cfg = dict(arena_size=400, num_pellets=400, num_viruses=6, mode=6)
    A = 6
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    seeds = np.arange(70, 70 + A).astype(np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for o, sd in zip(oras, seeds):
        o.seed(int(sd)); o.reset(True)
    rng = np.random.RandomState(8)
    for t in range(40):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = rng.randint(0, 3, size=(A, 1)).astype(np.int32)
        eng.set_actions(dxdy, act); eng.step()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); oras[a].step()
        if t % 7 == 3:
            eng.screen_obs(32, 32)               # another user of the staging buffer
        if t % 11 == 5:
            eng.gobigger_obs(64)
        got = eng.grid_obs(64, True, True, True, True)
        for a in range(A):
            assert np.array_equal(got[a, 0], oras[a].grid_obs(0, 64, True, True, True, True)), (t, a)
    eng.close()
