"""The batched RL surface (agarcl_amd/vector_env.py AgarioVectorEnv + VecEnvironment's persistent observation tensors): N arenas behind one
reset() / step() pair must be N single environments -- the `agarcl.GridEnvironment` objects gym.make("agario-grid-v0") drives
(/root/reference/gym_agario/AgarioEnv.py:85-132) -- step for step, including the episodic cut-off and the reset that follows a finished
episode, and step() must not wait for the GPU."""
import time

import numpy as np
import pytest


def _single_envs(N, seed, kw):
    from agarcl_amd import agarcl
    envs = []
    for a in range(N):
        e = agarcl.GridEnvironment(kw.get("num_agents", 1), kw.get("ticks_per_step", 4), kw["arena_size"], True, kw["num_pellets"], kw["num_viruses"], kw.get("num_bots", 0), 1, 0, kw.get("mode", 0))
        e.configure_observation(dict(grid_size=kw["grid_size"]))
        e.seed(seed + a); e.reset()
        envs.append(e)
    return envs


@pytest.mark.gpu
@pytest.mark.parametrize("sub_batches", [1, 3])
@pytest.mark.parametrize("kw", [dict(arena_size=200, num_pellets=300, num_viruses=4, grid_size=32, number_steps=7),
                                dict(arena_size=250, num_pellets=300, num_viruses=5, grid_size=16, mode=6, number_steps=11),
                                dict(arena_size=150, num_pellets=200, num_viruses=3, grid_size=16, num_bots=1, mode=8, number_steps=1000),
                                dict(arena_size=60, num_pellets=150, num_viruses=0, grid_size=16, num_bots=1, mode=9, num_agents=2, number_steps=1000)],
                         ids=["quiet", "mode6", "bot-mode8", "two-agents"])
def test_vector_env_equals_single_envs_step_for_step(hip_engine_cls, kw, sub_batches):
    """N arenas behind one step() -- as one launch (sub_batches=1) or as three sub-batches on streams of their own -- are N single envs.
    Several agents per arena: the arena is reset when ANY agent is done; the rows of agents that were not done are truncated."""
    import torch
    from agarcl_amd.vector_env import AgarioVectorEnv
    N, seed = 64, 900
    na = kw.get("num_agents", 1)
    steps = 40 if na == 1 else 200
    venv = AgarioVectorEnv(N, obs_type="grid", sub_batches=sub_batches, **kw)
    assert venv.concurrent_sub_batches == sub_batches and [c for _, c in venv.ranges] == ([64] if sub_batches == 1 else [22, 21, 21])
    obs, info = venv.reset(seed=seed)
    singles = _single_envs(N, seed, kw)
    want_shape = (N, 8, kw["grid_size"], kw["grid_size"]) if na == 1 else (N, na, 8, kw["grid_size"], kw["grid_size"])
    assert obs.shape == want_shape and obs.dtype == torch.int32 and obs.is_cuda and info == {}
    assert venv.single_observation_space.shape == want_shape[1:] and venv.observation_space.shape == want_shape
    got = obs.cpu().numpy().reshape(N, na, 8, kw["grid_size"], kw["grid_size"])
    for a, e in enumerate(singles):
        assert np.array_equal(got[a], np.stack(e.get_state())), "reset observation of arena %d" % a
    rng = np.random.RandomState(3)
    played = np.zeros(N, np.int64); resets = truncs = 0
    ret = np.zeros((N, na), np.float32); fin_ret = np.zeros((N, na), np.float32); fin_len = np.zeros(N, np.int64)   # the episode statistics, redone on the host
    for t in range(steps):
        move = rng.uniform(-1, 1, size=(N, na, 2)).astype(np.float32); kind = rng.randint(0, 3, size=(N, na)).astype(np.int32)
        shape = (lambda x: x[:, 0]) if na == 1 else (lambda x: x)
        obs, rew, term, trunc, info = venv.step((torch.as_tensor(shape(move), device="cuda"), torch.as_tensor(shape(kind), device="cuda")))
        assert rew.shape == want_shape[:1 + (na > 1)] and term.shape == rew.shape and term.dtype == torch.bool and trunc.dtype == torch.bool
        got = obs.cpu().numpy().reshape(N, na, 8, kw["grid_size"], kw["grid_size"])
        r, d, tr = (x.cpu().numpy().reshape(N, na) for x in (rew, term, trunc))
        ended = np.zeros(N, bool)
        for a, e in enumerate(singles):
            e.take_actions([(float(move[a, i, 0]), float(move[a, i, 1]), int(kind[a, i])) for i in range(na)])
            want_r = np.asarray(e.step(), np.float32)
            cut = played[a] >= kw["number_steps"]                                   # AgarioEnv.py:111-112, compared before the step is counted
            want_d = np.asarray(e.dones(), bool) | cut
            played[a] += 1
            assert np.array_equal(r[a], want_r) and np.array_equal(d[a], want_d), (t, a, r[a], want_r, d[a], want_d)
            ret[a] = (ret[a] + want_r).astype(np.float32)
            ended[a] = want_d.any()
            assert np.array_equal(tr[a], ended[a] & ~want_d), (t, a)
            truncs += int(tr[a].sum())
            if ended[a]:                                                            # what a user of the single env does next
                fin_ret[a], fin_len[a] = ret[a], played[a]
                e.reset(); played[a] = 0; resets += 1; ret[a] = 0
            assert np.array_equal(got[a], np.stack(e.get_state())), "step %d arena %d" % (t, a)
        assert np.array_equal(info["episode_steps"].cpu().numpy(), played)
        assert np.array_equal(info["episode_return"].cpu().numpy().reshape(N, na), ret) and np.array_equal(info["ended"].cpu().numpy(), ended)
        assert np.array_equal(info["final_return"].cpu().numpy().reshape(N, na), fin_ret) and np.array_equal(info["final_length"].cpu().numpy(), fin_len)
    assert resets >= (N if kw["number_steps"] < steps else 0)                     # the auto-reset path ran
    if na > 1:
        assert resets > 0 and truncs > 0, "the truncated-survivor path did not run (%d resets, %d truncations)" % (resets, truncs)
    for e in singles:
        e.close()
    venv.close()


@pytest.mark.gpu
def test_vector_env_send_recv_halves(hip_engine_cls):
    """double-buffered sampling: recv(j) / send(actions_j, j) per sub-batch, in an order of the caller's choosing, gives every arena the
    trajectory the full-batch step() gives it"""
    import torch
    from agarcl_amd.vector_env import AgarioVectorEnv
    N, kw = 96, dict(arena_size=200, num_pellets=300, num_viruses=4, mode=6, number_steps=9, k_pellets=4, k_viruses=2, k_others=2, k_cells=4)
    full = AgarioVectorEnv(N, obs_type="ram", **kw); halves = AgarioVectorEnv(N, obs_type="ram", sub_batches=2, **kw)
    obs, _ = full.reset(seed=31); halves.async_reset(seed=31)
    assert torch.equal(torch.cat([halves.recv(j)[0] for j in range(2)]), obs)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for t in range(30):
        move = torch.rand((N, 2), generator=g, device="cuda") * 2 - 1; kind = torch.randint(0, 3, (N,), generator=g, device="cuda", dtype=torch.int32)
        want = full.step((move, kind))
        for j in ((0, 1) if t % 2 else (1, 0)):
            lo, cnt = halves.ranges[j]
            halves.send((move[lo:lo + cnt], kind[lo:lo + cnt]), j)
        got = [halves.recv(j) for j in range(2)]
        for i in range(4):
            assert torch.equal(torch.cat([g_[i] for g_ in got]), want[i]), (t, i)
        for key in want[4]:
            assert torch.equal(torch.cat([g_[4][key] for g_ in got]), want[4][key]), (t, key)
    full.close(); halves.close()


def test_vector_env_spaces_without_a_gpu():
    """the spaces an RL library reads (gymnasium.vector.VectorEnv's attributes; the reference's per-env spaces are AgarioEnv.py:55-62, 232-264)
    are pure functions of the options: checked here through agarcl_amd/spaces.py without creating an engine"""
    from agarcl_amd import spaces
    sa = spaces.single_action_space(1, False)
    assert sa[0].shape == (2,) and float(sa[0].low.min()) == -1.0 and float(sa[0].high.max()) == 1.0 and sa[1].n == 3
    ba = spaces.batched_action_space(4096, 1, False)
    assert ba[0].shape == (4096, 2) and ba[1].shape == (4096,) and int(ba[1].nvec.max()) == 3
    assert spaces.batched_action_space(8, 3, True)[0].shape == (8, 3, 2)
    g = spaces.observation_space("grid", (128, 128, 8))
    assert g.shape == (128, 128, 8) and g.dtype == np.int32 and int(g.low.min()) == -1 and int(g.high.max()) == np.iinfo(np.int32).max
    s = spaces.observation_space("screen", (84, 84, 3))
    assert s.dtype == np.uint8 and int(s.high.max()) == 255 and s.contains(s.sample())
    assert sa.contains(sa.sample()) and ba.contains(ba.sample())


@pytest.mark.gpu
@pytest.mark.parametrize("obs_type,kw,shape,dtype", [("grid", dict(grid_size=32), (8, 32, 32), "int32"), ("grid", dict(grid_size=16, observe_pellets=False, channels_last=True), (16, 16, 6), "int32"),
                                                     ("screen", dict(screen_len=48), (48, 48, 3), "uint8"), ("screen", dict(screen_len=32, agent_view=True), (32, 32, 4), "uint8"),
                                                     ("ram", dict(k_cells=4, k_pellets=8, k_viruses=2, k_others=2), (4 + 12 + 16 + 6 + 6,), "float32")])
def test_vector_env_spaces_describe_what_step_returns(hip_engine_cls, obs_type, kw, shape, dtype):
    import torch
    from agarcl_amd.vector_env import AgarioVectorEnv
    N = 32
    venv = AgarioVectorEnv(N, obs_type=obs_type, arena_size=200, num_pellets=200, num_viruses=3, **kw)
    assert venv.num_envs == N and venv.single_observation_space.shape == shape and venv.observation_space.shape == (N,) + shape
    assert str(venv.single_observation_space.dtype) == dtype
    assert venv.single_action_space[0].shape == (2,) and venv.single_action_space[1].n == 3 and venv.action_space[0].shape == (N, 2)
    obs, _ = venv.reset(seed=2)
    move, kind = venv.action_space.sample()
    obs, rew, term, trunc, info = venv.step((np.asarray(move, np.float32), np.asarray(kind, np.int32)))     # host actions are uploaded
    assert tuple(obs.shape) == (N,) + shape and str(obs.dtype) == "torch." + dtype and rew.shape == (N,)
    lo, hi = float(np.min(venv.single_observation_space.low)), float(np.max(venv.single_observation_space.high))
    o = obs.double()
    assert float(o.min()) >= lo and float(o.max()) <= hi and float(o.abs().sum()) > 0
    venv.close()


@pytest.mark.gpu
def test_vector_env_step_does_not_wait_for_the_gpu(hip_engine_cls):
    """step() only enqueues: with ~0.3 s of GPU work already queued in front of it, the call returns in a few milliseconds, and the work it
    enqueued is still correct afterwards."""
    import torch
    from agarcl_amd.vector_env import AgarioVectorEnv
    N = 256
    venv = AgarioVectorEnv(N, obs_type="grid", arena_size=300, num_pellets=300, num_viruses=3, grid_size=32, number_steps=5)
    ref = AgarioVectorEnv(N, obs_type="grid", arena_size=300, num_pellets=300, num_viruses=3, grid_size=32, number_steps=5)
    venv.reset(seed=5); ref.reset(seed=5)
    move = torch.rand((N, 2), device="cuda") * 2 - 1; kind = torch.zeros(N, dtype=torch.int32, device="cuda")
    for _ in range(3):
        venv.step((move, kind)); ref.step((move, kind))        # warm: observation tensors exist, kernels are loaded
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(int(0.3 * 2.0e9))                         # ~0.3 s of GPU time in front of the step (cycles of a >= 1 GHz clock)
    e1.record()
    t0 = time.perf_counter()
    for _ in range(8):                                          # incl. the auto-reset after the cut-off
        out = venv.step((move, kind))
    host_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    queued_s = e0.elapsed_time(e1) * 1e-3
    assert queued_s > 0.05, queued_s
    assert host_s < 0.5 * queued_s, "8 steps took %.3f s on the host with %.3f s of GPU work queued in front: something waited" % (host_s, queued_s)
    for _ in range(8):
        want = ref.step((move, kind))
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1]) and torch.equal(out[2], want[2])
    venv.close(); ref.close()


@pytest.mark.gpu
def test_vec_environment_observation_tensors(hip_engine_cls):
    """VecEnvironment.grid_obs / screen_obs / gobigger_obs / ram_obs: persistent CUDA tensors equal to the engine's host copies"""
    import torch
    from agarcl_amd.vec_env import VecEnvironment
    A = 32
    env = VecEnvironment(A, num_agents=1, arena_size=300, num_pellets=300, num_viruses=5, num_bots=2, mode_number=0)
    env.seed(base_seed=77); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    first = {}
    for t in range(6):
        env.take_actions(torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32))
        env.step()
        grid = env.grid_obs(32); scr = env.screen_obs(48, 48); ram = env.ram_obs(); gb = env.gobigger_obs(32)
        for k, v in (("grid", grid), ("screen", scr), ("ram", ram), ("hdr", gb["hdr"])):
            assert first.setdefault(k, v.data_ptr()) == v.data_ptr()        # the same tensor every step: nothing is allocated in the loop
        assert np.array_equal(grid.cpu().numpy(), env.engine.grid_obs(32))
        assert np.array_equal(scr.cpu().numpy(), env.engine.screen_obs(48, 48))
        assert np.array_equal(ram.cpu().numpy(), env.engine.ram_obs(), equal_nan=True)
        host = env.engine.gobigger_obs(32)
        for k in host:
            assert np.array_equal(gb[k].cpu().numpy(), host[k]), k
    other = torch.empty_like(grid)
    assert env.grid_obs(32, out=other) is other and torch.equal(other, env.grid_obs(32))
    with pytest.raises(ValueError):
        env.grid_obs(32, out=torch.empty((A, 1, 8, 16, 16), dtype=torch.int32, device="cuda"))
    env.close()


@pytest.mark.gpu
def test_rollout_example_runs(hip_engine_cls):
    """examples/vector_rollout.py end to end on a small batch (its own process, as a user would run it): with short episodes the auto-reset
    and the episode statistics are exercised, and the bare mode times step() alone"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "vector_rollout.py"), "--envs", "64", "--steps", "30", "--obs", "ram", "--number-steps", "10"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    assert "env-steps/s through AgarioVectorEnv.step" in line and " episodes ended" in line
    assert int(line.split(";")[1].split()[0]) >= 64 * 2      # 30 steps of 10-step episodes: every arena ended at least twice
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "vector_rollout.py"), "--envs", "64", "--steps", "10", "--obs", "none", "--bare"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "step() alone" in out.stdout, out.stderr[-2000:]
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "vector_rollout.py"), "--envs", "96", "--steps", "25", "--obs", "ram", "--number-steps", "10",
                          "--sub-batches", "2", "--halves", "--mode", "6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "as 2 ranges (recv / send halves)" in out.stdout, out.stderr[-2000:]
    assert int(out.stdout.strip().splitlines()[-1].split(";")[1].split()[0]) >= 96      # every arena ended at least once


@pytest.mark.gpu
def test_vector_env_defaults_and_flagged_arenas_gpu(hip_engine_cls):
    """(1) sub_batches="auto": ONE range for the full-batch step() whatever the workload, vec_env.default_sub_batches with halves=True (round 6: the
    measurement beside the choice in vector_env.py).  (2) on_capacity_flag="reset": an arena that raises a capacity flag (two food slots, an agent that
    keeps ejecting) is cut inside the step like an ended episode -- truncated, not terminated, its flag gone -- where the default raises."""
    import torch
    from agarcl_amd import _capi
    from agarcl_amd.vector_env import AgarioVectorEnv
    v = AgarioVectorEnv(1024, obs_type="none", mode=6, num_viruses=5); assert v.sub_batches == 1 and v.pipe is None; v.close()
    v = AgarioVectorEnv(1024, obs_type="none", mode=6, num_viruses=5, halves=True); assert v.sub_batches == 4 and len(v.ranges) == 4; v.close()
    v = AgarioVectorEnv(1024, obs_type="none", mode=0, halves=True); assert v.sub_batches == 1; v.close()
    N = 64
    kw = dict(obs_type="none", arena_size=120, num_pellets=100, num_viruses=0, mode=6, cap_foods=2, number_steps=100000)
    move = torch.zeros((N, 2), device="cuda"); move[:, 0] = 1.0; feed = torch.ones(N, dtype=torch.int32, device="cuda")
    v = AgarioVectorEnv(N, on_capacity_flag="reset", **kw); v.reset(seed=5)
    cut = 0
    for t in range(40):
        obs, rew, term, trunc, info = v.step((move, feed))
        torch.cuda.synchronize()
        assert not bool(term.any()) and torch.equal(trunc, info["ended"])
        cut += int(info["ended"].sum())
        assert not v._parts[0].engine.flags().any(), "a flagged arena survived the step"
    assert cut >= N // 2, cut
    v.close()
    v = AgarioVectorEnv(N, **kw); v.reset(seed=5)          # the default: the watch raises (asynchronously: within ~64 steps)
    with pytest.raises(_capi.AgarclError):
        for t in range(200):
            v.step((move, feed))
    v.close()
