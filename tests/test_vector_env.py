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
        e = agarcl.GridEnvironment(1, kw.get("ticks_per_step", 4), kw["arena_size"], True, kw["num_pellets"], kw["num_viruses"], kw.get("num_bots", 0), 1, 0, kw.get("mode", 0))
        e.configure_observation(dict(grid_size=kw["grid_size"]))
        e.seed(seed + a); e.reset()
        envs.append(e)
    return envs


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(arena_size=200, num_pellets=300, num_viruses=4, grid_size=32, number_steps=7),
                                dict(arena_size=250, num_pellets=300, num_viruses=5, grid_size=16, mode=6, number_steps=11),
                                dict(arena_size=150, num_pellets=200, num_viruses=3, grid_size=16, num_bots=1, mode=8, number_steps=1000)],
                         ids=["quiet", "mode6", "bot-mode8"])
def test_vector_env_equals_single_envs_step_for_step(hip_engine_cls, kw):
    import torch
    from agarcl_amd.vector_env import AgarioVectorEnv
    N, seed, steps = 64, 900, 40
    venv = AgarioVectorEnv(N, obs_type="grid", **kw)
    obs, info = venv.reset(seed=seed)
    singles = _single_envs(N, seed, kw)
    assert obs.shape == (N, 8, kw["grid_size"], kw["grid_size"]) and obs.dtype == torch.int32 and obs.is_cuda and info == {}
    got = obs.cpu().numpy()
    for a, e in enumerate(singles):
        assert np.array_equal(got[a], e.get_state()[0]), "reset observation of arena %d" % a
    rng = np.random.RandomState(3)
    played = np.zeros(N, np.int64); resets = 0
    ret = np.zeros(N, np.float32); fin_ret = np.zeros(N, np.float32); fin_len = np.zeros(N, np.int64)   # the episode statistics, redone on the host
    for t in range(steps):
        move = rng.uniform(-1, 1, size=(N, 2)).astype(np.float32); kind = rng.randint(0, 3, size=N).astype(np.int32)
        obs, rew, term, trunc, info = venv.step((torch.as_tensor(move, device="cuda"), torch.as_tensor(kind, device="cuda")))
        got, r, d = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        assert rew.shape == (N,) and term.shape == (N,) and term.dtype == torch.bool and not trunc.any().item()
        for a, e in enumerate(singles):
            e.take_actions([(float(move[a, 0]), float(move[a, 1]), int(kind[a]))])
            want_r = e.step()[0]
            want_d = bool(e.dones()[0]) or played[a] >= kw["number_steps"]        # AgarioEnv.py:111-112, compared before the step is counted
            played[a] += 1
            assert r[a] == np.float32(want_r) and bool(d[a]) == want_d, (t, a, r[a], want_r, d[a], want_d)
            ret[a] = np.float32(ret[a] + np.float32(want_r))
            if want_d:                                                            # what a user of the single env does next
                fin_ret[a], fin_len[a] = ret[a], played[a]
                e.reset(); played[a] = 0; resets += 1; ret[a] = 0
            assert np.array_equal(got[a], e.get_state()[0]), "step %d arena %d" % (t, a)
        assert np.array_equal(info["episode_steps"].cpu().numpy(), played)
        assert np.array_equal(info["episode_return"].cpu().numpy(), ret) and np.array_equal(info["ended"].cpu().numpy(), d)
        assert np.array_equal(info["final_return"].cpu().numpy(), fin_ret) and np.array_equal(info["final_length"].cpu().numpy(), fin_len)
    assert resets >= (N if kw["number_steps"] < steps else 0)                     # the auto-reset path ran
    for e in singles:
        e.close()
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
