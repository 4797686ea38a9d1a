"""Shared lock-step comparison helpers (TEST INFRASTRUCTURE)."""
import numpy as np
from oracle import blob


def policy(seed, t, n_agents, allow_actions=True, sticky=1):
    """Deterministic pseudo-random policy: (dxdy[n,2] float32, act[n] int32) for decision step t."""
    rng = np.random.RandomState((seed * 1000003 + (t // sticky)) & 0x7FFFFFFF)
    dxdy = rng.uniform(-1, 1, size=(n_agents, 2)).astype(np.float32)
    act = rng.randint(0, 3, size=n_agents).astype(np.int32) if allow_actions else np.zeros(n_agents, np.int32)
    return dxdy, act


def run_lockstep(envs, ticks, seed, policy_seed=1, allow_actions=True, every=1, sticky=1, act_every=4,
                 respawn=True, rtol=0.0, on_tick=None):
    """Drive all `envs` (objects with RefEnv's engine-level API) with the same inputs; compare the
    first env's blob with every other env's blob every `every` ticks.  Returns (ok, message)."""
    for e in envs:
        e.seed(seed)
        e.reset(True)
    b0 = envs[0].dump()
    for k, e in enumerate(envs[1:], 1):
        d = blob.diff(b0, e.dump(), rtol)
        if d:
            return False, "after reset: env%d: %s" % (k, d)
    n_agents = envs[0].num_agents
    pids = envs[0].pids()
    for t in range(ticks):
        if t % act_every == 0:
            dxdy, act = policy(policy_seed, t // act_every, n_agents, allow_actions, sticky)
            for e in envs:
                for i, pid in enumerate(pids):
                    e.take_action(pid, float(dxdy[i, 0]), float(dxdy[i, 1]), int(act[i]))
        for e in envs:
            e.tick()
        if respawn and t % act_every == act_every - 1:
            for e in envs:
                e.respawn_dead()
        if on_tick:
            on_tick(t, envs)
        if (t + 1) % every == 0 or t == ticks - 1:
            b0 = envs[0].dump()
            for k, e in enumerate(envs[1:], 1):
                d = blob.diff(b0, e.dump(), rtol)
                if d:
                    return False, "tick %d: env%d: %s" % (t, k, d)
    return True, "ok"
