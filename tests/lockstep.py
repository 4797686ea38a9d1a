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


def run_batched_lockstep(engine, oracles, steps, seeds, policy_seed=1, allow_actions=True, sticky=1,
                         ticks_per_step=4, every=1, rtol=0.0, compare_rewards=True):
    """engine: agarcl_amd._capi.BatchedEngine with len(oracles) arenas; oracles: OraEnv/RefEnv list
    (one agent each).  Same per-arena seeds and per-arena random actions; env-level stepping
    (take_actions + step).  Compares every arena's blob (and rewards/dones) every `every` steps."""
    A = len(oracles)
    engine.seed(np.asarray(seeds, dtype=np.uint32))
    engine.reset(reset_ids=True)
    for o, s in zip(oracles, seeds):
        o.seed(int(s))
        o.reset(True)
    for a in range(A):
        d = blob.diff(oracles[a].dump(), engine.dump(a), rtol)
        if d:
            return False, "after reset: arena %d: %s" % (a, d)
    for t in range(steps):
        dxdy = np.zeros((A, 1, 2), np.float32)
        act = np.zeros((A, 1), np.int32)
        for a in range(A):
            dd, aa = policy(policy_seed + 7919 * a, t, 1, allow_actions, sticky)
            dxdy[a, 0] = dd[0]; act[a, 0] = aa[0]
        engine.set_actions(dxdy, act)
        engine.step(ticks_per_step)
        rw = []
        for a in range(A):
            oracles[a].take_actions(dxdy[a], act[a])
            rw.append(oracles[a].step())
        if (t + 1) % every == 0 or t == steps - 1:
            fl = engine.flags()
            if fl.any():
                return False, "step %d: capacity flags raised %s" % (t, fl.tolist())
            for a in range(A):
                d = blob.diff(oracles[a].dump(), engine.dump(a), rtol)
                if d:
                    return False, "step %d: arena %d: %s" % (t, a, d)
            if compare_rewards:
                er = engine.rewards()
                for a in range(A):
                    if not np.array_equal(er[a], rw[a]):
                        return False, "step %d: arena %d: rewards %s vs %s" % (t, a, er[a], rw[a])
                ed = engine.dones()
                for a in range(A):
                    if not np.array_equal(ed[a], oracles[a].dones()):
                        return False, "step %d: arena %d: dones differ" % (t, a)
    return True, "ok"
