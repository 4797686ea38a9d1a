"""Screen observation.  Rule-level parity only and UNPINNED (no OpenGL here): the HIP
rasteriser is compared with the numpy restatement of the same rules (oracle/screen_oracle.py) within a tolerance on
polygon-edge pixels, plus structural properties of the frame; the respawn hook is compared with the oracle."""
import numpy as np
import pytest

from lockstep import run_batched_lockstep


def test_screen_respawn_hook_on_emulated_kernels(emu_lib, oracle_lib):
    """E5: with the hook a dead agent is respawned after the ticks in a mode that never respawns otherwise, and that
    step's reward carries + c_death (BaseEnvironment.hpp:116-120)."""
    from agarcl_amd import _capi
    cfg = dict(num_agents=2, arena_size=120, num_pellets=150, num_viruses=2, num_bots=0, mode=4, c_death=-50)
    A = 4
    eng = _capi.BatchedEngine(A, lib=emu_lib, screen_respawn=True, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    for o in oras:
        o.set_screen_hook(True)
    ok, msg = run_batched_lockstep(eng, oras, 700, seeds=np.arange(300, 300 + A), sticky=6, every=5)
    assert ok, msg
    eng.close()


@pytest.mark.gpu
def test_screen_respawn_hook_hip(hip_engine_cls, oracle_lib):
    """E5 on the GPU: ScreenEnvironment's hook (ScreenEnvironment.hpp:233-243) in lock-step with the oracle, in a mode that
    never respawns otherwise (mode 4) and in mode 0 with bots, where BaseEnvironment's own respawn follows the hook."""
    for cfg, steps in ((dict(num_agents=2, arena_size=120, num_pellets=150, num_viruses=2, num_bots=0, mode=4, c_death=-50), 700),
                       (dict(num_agents=1, arena_size=150, num_pellets=200, num_viruses=3, num_bots=3, mode=0, c_death=-7), 500)):
        A = 8
        eng = hip_engine_cls(A, screen_respawn=True, **cfg)
        oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
        for o in oras:
            o.set_screen_hook(True)
        ok, msg = run_batched_lockstep(eng, oras, steps, seeds=np.arange(300, 300 + A), sticky=6, every=5)
        eng.close()
        assert ok, "%s: %s" % (cfg, msg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,steps", [
    (dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6), 40),
    (dict(num_agents=2, arena_size=200, num_pellets=300, num_viruses=5, num_bots=3, mode=0), 60),
    (dict(arena_size=60, num_pellets=100, num_viruses=2, mode=0), 10),      # arena smaller than the view: grid + outside
])
def test_hip_screen_frames_follow_the_rules(hip_engine_cls, cfg, steps):
    from oracle import screen_oracle
    A = 3
    na = cfg.get("num_agents", 1)
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 77); eng.reset(reset_ids=True)
    rng = np.random.RandomState(2)
    for t in range(steps):
        eng.set_actions(rng.uniform(-1, 1, size=(A, na, 2)).astype(np.float32), rng.randint(0, 3, size=(A, na)).astype(np.int32)); eng.step()
    for (W, H) in ((84, 84), (96, 64)):
        frames = eng.screen_obs(W, H)
        assert frames.shape == (A, na, H, W, 3) and frames.dtype == np.uint8
        for a in range(A):
            ar, pl = eng.arena_words(a)
            b = eng.dump(a)
            kinds = [int(pl[int(ar[13 + k]), 16]) for k in range(pl.shape[0])]      # PL_KIND in iteration order
            for i in range(na):
                ref = screen_oracle.render(b, cfg["arena_size"], int(pl[i, 15]), kinds, W, H)   # PL_PID of agent slot i
                diff = (ref != frames[a, i]).any(axis=2).mean()
                assert diff <= 0.004, "arena %d agent %d %dx%d: %.2f%% of the pixels differ" % (a, i, W, H, 100 * diff)
                assert frames[a, i].reshape(-1, 3).max() == 255
    # agent view (N4): 4 channels by type + the reference's byte-level post-processing
    W = H = 84
    frames = eng.screen_obs(W, H, agent_view=True)
    assert frames.shape == (A, na, H, W, 4)
    for a in range(A):
        ar, pl = eng.arena_words(a)
        b = eng.dump(a)
        kinds = [int(pl[int(ar[13 + k]), 16]) for k in range(pl.shape[0])]
        for i in range(na):
            ref = screen_oracle.render(b, cfg["arena_size"], int(pl[i, 15]), kinds, W, H, agent_view=True, main_pid=int(pl[na - 1, 15]))
            diff = (ref != frames[a, i]).any(axis=2).mean()
            assert diff <= 0.006, "agent view, arena %d agent %d: %.2f%% of the pixels differ" % (a, i, 100 * diff)
            f = frames[a, i]
            assert set(np.unique(f[:, :, 0])) <= {0, 255} and set(np.unique(f[:, :, 1])) <= {0, 255} and set(np.unique(f[:, :, 2])) <= {0, 255}
            assert (f[:, :, 3] == 230).any() or i != na - 1       # the main agent lives in the alpha channel
    eng.close()


@pytest.mark.gpu
def test_agarcl_screen_environment_mirror():
    from agarcl_amd import agarcl
    assert agarcl.has_screen_env
    env = agarcl.ScreenEnvironment(1, 4, 1000, True, 1000, 25, 0, True, 0, 6, False, 84, 84, False)
    env.seed(3); env.reset()
    for t in range(10):
        env.take_actions([(0.3, -0.2, 0)]); env.step()
    assert env.observation_shape() == (1, 84, 84, 3)
    s = env.get_state()
    assert isinstance(s, list) and len(s) == 1            # bindings.cpp:157-168: a list holding the one frame buffer
    s = s[0]
    assert s.shape == (1, 84, 84, 3) and s.dtype == np.uint8
    img = s.reshape(84, 84, 3)
    assert (img == 255).all(axis=2).mean() > 0.3            # white background dominates
    assert not (img[40:44, 40:44] == 255).all()             # the agent's cells sit at the view centre
    env.close()
    env = agarcl.ScreenEnvironment(1, 4, 1000, True, 1000, 25, 0, True, 0, 6, False, 84, 84, True)   # agent view: 4 channels
    env.seed(3); env.reset(); env.take_actions([(0.3, -0.2, 0)]); env.step()
    assert env.observation_shape() == (1, 84, 84, 4) and env.get_state()[0].shape == (1, 84, 84, 4)
    env.close()


@pytest.mark.gpu
def test_gym_wrapper_screen_observation():
    """AgarioEnv(obs_type="screen") as gym_agario builds it (AgarioEnv.py:235-250): (1, W, H, 3) uint8 frames."""
    from agarcl_amd import gym_agario
    g = gym_agario.AgarioEnv(obs_type="screen", difficulty="normal", screen_len=64, number_steps=4)
    g.seed(2)
    obs, info = g.reset()
    assert obs.shape == (1, 64, 64, 3)       # the reference yields the (num_frames, W, H, C) buffer as is (AgarioEnv.py:196-197)
    assert obs.dtype == np.uint8 and info == {}
    assert g.observation_shape == (1, 64, 64, 3)
    for k in range(3):
        obs, rew, done, trunc, info = g.step(((0.2, -0.4), 0))
        assert obs.shape == (1, 64, 64, 3) and obs.dtype == np.uint8 and isinstance(rew, float) and trunc is False
    g.close()


@pytest.mark.gpu
def test_screen_obs_device_buffer_4096(hip_engine_cls):
    """BASELINE config 5's other half at its full size: uint8 [4096][1][84][84][3] written straight into a torch HBM tensor, on the
    full rule set after 40 steps (agents split, eject, pop on viruses).  A sample of arenas is compared with the host restatement of the
    rasterisation rules (tolerance on polygon edges, as above); ALL frames through size-independent properties."""
    import torch
    from oracle import screen_oracle
    A, W, H = 4096, 84, 84
    cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(3)
    acts = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(8)]
    mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
    for t in range(40):
        eng.set_actions(mv[t % 8], acts[t % 8]); eng.step()
    out = torch.full((A, 1, H, W, 3), 7, dtype=torch.uint8, device="cuda")
    eng.screen_obs(W, H, out_ptr=out.data_ptr()); eng.sync()
    f = out.cpu().numpy()
    again = torch.empty_like(out); eng.screen_obs(W, H, out_ptr=again.data_ptr()); eng.sync()
    assert torch.equal(out, again)                                              # deterministic
    assert np.array_equal(f[:16], eng.screen_obs(W, H)[:16])                    # the host path returns the same frames
    for a in list(range(0, A, 293)) + [A - 1]:                                  # 15 arenas against the rule restatement
        ar, pl = eng.arena_words(a)
        ref = screen_oracle.render(eng.dump(a), cfg["arena_size"], int(pl[0, 15]), [0], W, H)
        diff = (ref != f[a, 0]).any(axis=2).mean()
        assert diff <= 0.004, "arena %d: %.2f%% of the pixels differ" % (a, 100 * diff)
    flat = f.reshape(A, H * W, 3)
    white = (flat == 255).all(axis=2).mean(axis=1)
    assert (flat.max(axis=(1, 2)) == 255).all()                                 # every frame has background
    assert white.min() > 0.05 and white.mean() > 0.3                            # ... and it dominates on average (a mass-1000 agent's cells cover the view centre)
    centre = f[:, 0, H // 2 - 1:H // 2 + 1, W // 2 - 1:W // 2 + 1].reshape(A, -1, 3)
    assert (~(centre == 255).all(axis=2)).any(axis=1).mean() > 0.95             # the camera sits on the agent: something is drawn at the centre
    # agent view (N4) at the full size: three binary channels + the main agent in the fourth
    av = torch.empty((A, 1, H, W, 4), dtype=torch.uint8, device="cuda")
    eng.screen_obs(W, H, out_ptr=av.data_ptr(), agent_view=True); eng.sync()
    g = av.cpu().numpy()
    assert set(np.unique(g[..., :3])) <= {0, 255}
    assert ((g[..., 3] == 230).reshape(A, -1).any(axis=1)).mean() > 0.95
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,steps", [
    (dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6), 40),
    (dict(num_agents=2, arena_size=200, num_pellets=300, num_viruses=5, num_bots=3, mode=0), 60),
    (dict(arena_size=60, num_pellets=100, num_viruses=2, mode=0), 10),
    # the paper's task 1 (squared pellet pattern: the view full of pellets, runs of 255-pixels one pixel apart -- the run pass's run-by-run path -- and
    # next to each other across chunk boundaries) and a dense small arena (several hundred listed pellets: every wavefront's share of the list)
    (dict(arena_size=350, num_pellets=500, num_viruses=0, mode=1), 30),
    (dict(arena_size=80, num_pellets=1300, num_viruses=0, mode=6), 20),
])
def test_band_rasteriser_equals_the_pixelwise_kernel(hip_engine_cls, monkeypatch, cfg, steps):
    """k_screen_obs paints bounding boxes into an LDS band; k_screen_obs_pixelwise shades every pixel against every entity: the same rules,
    so the same bytes -- at the training size, a non-square size, the 512 x 512 frame of get_frame() (several bands) and in agent view."""
    A = 5
    na = cfg.get("num_agents", 1)
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 91); eng.reset(reset_ids=True)
    rng = np.random.RandomState(6)
    for t in range(steps):
        eng.set_actions(rng.uniform(-1, 1, size=(A, na, 2)).astype(np.float32), rng.randint(0, 3, size=(A, na)).astype(np.int32)); eng.step()
    for (W, H, av) in ((84, 84, False), (96, 64, False), (512, 512, False), (84, 84, True), (200, 120, True), (1024, 33, False),
                       (128, 128, True), (50, 37, False), (77, 31, False), (33, 45, True)):   # (the paper's task frame; byte counts that leave frames / bands off 4-byte boundaries: the word-wide and the byte-wise output loops)
        monkeypatch.delenv("AGARCL_SCREEN_PIXELWISE", raising=False)
        new = eng.screen_obs(W, H, agent_view=av)
        monkeypatch.setenv("AGARCL_SCREEN_PIXELWISE", "1")
        old = eng.screen_obs(W, H, agent_view=av)
        assert np.array_equal(new, old), (W, H, av, float((new != old).mean()))
    eng.close()


@pytest.mark.gpu
def test_band_rasteriser_tiny_and_odd_frames(hip_engine_cls, monkeypatch):
    """frames smaller than a wavefront's share -- 1 x 1, a single row, a single column, fewer rows than wavefronts -- and widths that are not a
    multiple of four: the per-wavefront band loop, its look-back across wavefront boundaries (agent view) and the byte-wise output loops, against the
    pixel-wise kernel"""
    A = 6
    eng = hip_engine_cls(A, arena_size=120, num_pellets=400, num_viruses=3, mode=6)
    eng.seed(None, 17); eng.reset(reset_ids=True)
    rng = np.random.RandomState(2)
    for t in range(25):
        eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32)); eng.step()
    for (W, H) in ((1, 1), (2, 1), (1, 7), (9, 1), (3, 2), (2, 5), (5, 3), (7, 7), (130, 3), (3, 130), (257, 5), (64, 64), (66, 9)):
        for av in (False, True):
            monkeypatch.delenv("AGARCL_SCREEN_PIXELWISE", raising=False)
            new = eng.screen_obs(W, H, agent_view=av)
            monkeypatch.setenv("AGARCL_SCREEN_PIXELWISE", "1")
            old = eng.screen_obs(W, H, agent_view=av)
            assert np.array_equal(new, old), (W, H, av, float((new != old).mean()))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(arena_size=350, num_pellets=500, num_viruses=0, mode=6), dict(arena_size=350, num_pellets=500, num_viruses=0, mode=3)])
def test_band_rasteriser_many_frames_tight_boxes(hip_engine_cls, monkeypatch, cfg):
    """round 6: entity boxes tight to 1/32 of a pixel (a far pellet covers one pixel centre or none), runs of alike hits painted a lane each, the look-back
    pixels only on demand -- 768 frames of the paper's task frame and of the 84 x 84 frame at two points of the game against the pixel-wise kernel"""
    A = 768
    eng = hip_engine_cls(A, **cfg)
    eng.seed(None, 4242); eng.reset(reset_ids=True)
    rng = np.random.RandomState(11)
    for steps in (5, 50):
        for t in range(steps):
            eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32)); eng.step()
        for (W, H, av) in ((128, 128, True), (84, 84, False), (84, 84, True), (128, 128, False)):
            monkeypatch.delenv("AGARCL_SCREEN_PIXELWISE", raising=False)
            new = eng.screen_obs(W, H, agent_view=av)
            monkeypatch.setenv("AGARCL_SCREEN_PIXELWISE", "1")
            old = eng.screen_obs(W, H, agent_view=av)
            bad = (new.reshape(A, -1) != old.reshape(A, -1)).any(axis=1)
            assert not bad.any(), (W, H, av, int(bad.sum()))
    eng.close()
