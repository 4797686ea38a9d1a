"""The reference's ten RL tasks (/root/reference/bench/tasks_configs/mode_{1..10}.json: 350 x 350 arena, 500 pellets, the 128 x 128 agent-view
screen observation, number_steps 500 / 3000 / 10000, one bot in modes 7-10) -- the workloads the paper trains on -- as a suite on the batched
surface.  Their VALUES live in tests/golden/paper_tasks.json (written by tests/golden/make_tasks_fixture.py in the build container); goldens
recorded from the reference itself for modes 1, 4, 7, 10 (tests/golden/task_mode*.npz) are replayed by test_golden_oracle.py /
test_gpu_parity.py on the oracle, the emulation and the HIP engine.  Here every task runs through AgarioVectorEnv (one agarcl_vec_step per
step: engine step, episode bookkeeping, same-step auto-reset, the agent-view frame) in lock-step with the oracle driven the way a user of the
single env drives it (AgarioEnv.py:85-132)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import blob

FIXTURE = os.path.join(ROOT, "tests", "golden", "paper_tasks.json")
TASKS = json.load(open(FIXTURE))["tasks"]


def test_fixture_holds_the_ten_tasks():
    assert sorted(TASKS, key=int) == [str(m) for m in range(1, 11)]
    for m in range(1, 11):
        t = TASKS[str(m)]
        assert t["mode"] == m and t["arena_size"] == 350 and t["num_pellets"] == 500 and t["num_viruses"] == 0 and t["ticks_per_step"] == 4
        assert t["obs_type"] == "screen" and t["screen_len"] == 128 and t["agent_view"] is True and t["env_type"] == 0 and t["reward_type"] == 1
        assert t["num_bots"] == (1 if m >= 7 else 0) and t["number_steps"] == (500 if m <= 2 else 3000 if m <= 6 else 10000)


def _engine_cfg(t):
    return dict(num_agents=1, ticks_per_step=t["ticks_per_step"], arena_size=t["arena_size"], pellet_regen=bool(t["pellet_regen"]), num_pellets=t["num_pellets"],
                num_viruses=t["num_viruses"], num_bots=t["num_bots"], reward_type=t["reward_type"], c_death=t["c_death"], mode=t["mode"])


def _lockstep(venv, oras, steps, number_steps, seed, check_every, frames_every):
    """drives the vector env and one oracle per arena with the same actions; the oracle side does per arena what a user of the single env
    does: step, done = engine flag or the cut-off (compared before the step is counted), reset when done"""
    import torch
    N = len(oras)
    eng = venv.env.engine
    obs, _ = venv.reset(seed=seed)
    for a, o in enumerate(oras):
        o.set_screen_hook(True)           # obs_type "screen": ScreenEnvironment's respawn hook (ScreenEnvironment.hpp:233-243)
        o.seed(seed + a); o.reset(False)
    assert tuple(obs.shape) == (N, 128, 128, 4) and obs.dtype == torch.uint8
    for a in range(N):
        d = blob.diff(oras[a].dump(), eng.dump(a), 0.0)
        assert not d, "after reset, arena %d: %s" % (a, d)
    rng = np.random.RandomState(11)
    played = np.zeros(N, np.int64); ended_total = engine_done_total = 0
    move = np.zeros((N, 2), np.float32); kind = np.zeros(N, np.int32)
    for t in range(steps):
        if t % 6 == 0:                    # sticky actions: agents travel, eat, split, meet the bot
            move = rng.uniform(-1, 1, size=(N, 2)).astype(np.float32); kind = rng.randint(0, 3, size=N).astype(np.int32)
        obs, rew, term, trunc, info = venv.step((torch.as_tensor(move, device="cuda"), torch.as_tensor(kind, device="cuda")))
        r, d, ended = rew.cpu().numpy(), term.cpu().numpy(), info["ended"].cpu().numpy()
        assert not trunc.any().item()
        for a, o in enumerate(oras):
            o.take_actions(move[a], kind[a])
            want_r = np.float32(o.step()[0])
            flag = bool(o.dones()[0])
            want_d = flag or played[a] >= number_steps
            played[a] += 1
            assert r[a] == want_r and bool(d[a]) == want_d and bool(ended[a]) == want_d, (t, a, r[a], want_r, d[a], want_d)
            if want_d:
                o.reset(False); played[a] = 0; ended_total += 1; engine_done_total += int(flag)
        assert np.array_equal(info["episode_steps"].cpu().numpy(), played)
        if t % check_every == check_every - 1 or t == steps - 1:
            assert not eng.flags().any()
            for a in range(N):
                dd = blob.diff(oras[a].dump(), eng.dump(a), 0.0)
                assert not dd, "step %d arena %d: %s" % (t, a, dd)
        if t % frames_every == frames_every - 1:
            # the frame written inside the one-call step is the frame the observation entry point writes for the same state
            host = eng.screen_obs(128, 128, agent_view=True)[:, 0]
            got = obs.cpu().numpy()
            assert np.array_equal(got, host) and got.max() == 255 and (got != got[:, :1, :1]).any()
    return ended_total, engine_done_total


@pytest.mark.gpu
@pytest.mark.parametrize("m", range(1, 11))
def test_paper_task_in_lockstep_with_the_oracle(hip_engine_cls, oracle_lib, m):
    from agarcl_amd.vector_env import AgarioVectorEnv
    t = TASKS[str(m)]
    N = 16
    # (a) the task exactly as configured: 520 steps -- the cut-off of modes 1 and 2 (500 steps) falls inside, the others run on
    venv = AgarioVectorEnv(N, **t)
    assert venv.number_of_steps == t["number_steps"] and venv.single_observation_space.shape == (128, 128, 4)
    oras = [oracle_lib.OraEnv(**_engine_cfg(t)) for _ in range(N)]
    ended, by_engine = _lockstep(venv, oras, 520, t["number_steps"], 3000 + 100 * m, check_every=65, frames_every=130)
    if t["number_steps"] <= 500:
        assert ended >= N, "every arena reaches the cut-off at step %d" % t["number_steps"]
    venv.close()
    for o in oras:
        o.close()
    # (b) the same task with the cut-off shortened: the cut-off, the reset and the next episodes in every mode, 150 steps
    venv = AgarioVectorEnv(N, **dict(t, number_steps=37))
    oras = [oracle_lib.OraEnv(**_engine_cfg(t)) for _ in range(N)]
    ended, by_engine = _lockstep(venv, oras, 150, 37, 7000 + 100 * m, check_every=19, frames_every=75)
    assert ended >= 3 * N
    venv.close()
    for o in oras:
        o.close()


@pytest.mark.gpu
def test_paper_task_bot_episode_ends_by_the_engine(hip_engine_cls, oracle_lib):
    """modes 7-10 end when the bot or the agent dies (BaseEnvironment.hpp:104-113): in a small arena that happens soon -- the engine's own
    done flag, not the cut-off, drives the auto-reset here"""
    from agarcl_amd.vector_env import AgarioVectorEnv
    t = dict(TASKS["9"], arena_size=70, num_pellets=150)
    N = 16
    venv = AgarioVectorEnv(N, **t)
    oras = [oracle_lib.OraEnv(**_engine_cfg(t)) for _ in range(N)]
    ended, by_engine = _lockstep(venv, oras, 400, t["number_steps"], 5, check_every=40, frames_every=200)
    assert by_engine > 0, "no episode was ended by the engine's done flag"
    venv.close()
