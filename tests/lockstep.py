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
    n = engine.num_agents
    for t in range(steps):
        dxdy = np.zeros((A, n, 2), np.float32)
        act = np.zeros((A, n), np.int32)
        for a in range(A):
            dd, aa = policy(policy_seed + 7919 * a, t, n, allow_actions, sticky)
            dxdy[a] = dd; act[a] = aa
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


# ---- golden fixture replay -------------------------------------------------------------------------
class EngineAsEnv:
    """Adapts agarcl_amd._capi.BatchedEngine (one arena) to the RefEnv/OraEnv env-level API."""

    def __init__(self, engine_cls, lib=None, **cfg):
        kw = dict(cfg)
        self.num_agents = kw.get("num_agents", 1)
        self.e = engine_cls(1, lib=lib, **kw) if lib is not None else engine_cls(1, **kw)

    def seed(self, s):
        self.e.seed(np.asarray([s], dtype=np.uint32))

    def reset(self, reset_ids=True):
        self.e.reset(reset_ids=reset_ids)

    def take_actions(self, dxdy, act):
        self.e.set_actions(np.asarray(dxdy, np.float32).reshape(1, self.num_agents, 2), np.asarray(act, np.int32).reshape(1, self.num_agents))

    def step(self):
        self.e.step(0)
        return self.e.rewards()[0]

    def dones(self):
        return self.e.dones()[0]

    def dump(self):
        return self.e.dump(0)

    def load(self, b):
        self.e.load(b, 0)

    def flags(self):
        return int(self.e.flags()[0])

    # engine-level driving (Engine::tick without BaseEnvironment::step around it), as RefEnv / OraEnv offer it
    def pids(self):
        ar, pl = self.e.arena_words(0)
        return [int(pl[int(ar[13 + k]), 15]) for k in range(self.e.players) if int(pl[int(ar[13 + k]), 16]) == 0]   # AR_ORDER0, PL_PID, PL_KIND

    def set_player(self, pid, tx, ty, action):
        _, pl = self.e.arena_words(0)
        txy = np.zeros((1, self.e.players, 2), np.float32); act = np.zeros((1, self.e.players), np.int32)
        for slot in range(self.e.players):
            txy[0, slot] = pl[slot, 2:4].view(np.float32); act[0, slot] = pl[slot, 1]
            if int(pl[slot, 15]) == pid:
                txy[0, slot] = (tx, ty); act[0, slot] = action
        self.e.set_targets(txy, act)

    def tick(self):
        self.e.tick(1)


def golden_crc(b):
    import zlib
    f = b.view(np.float32)
    c = b.copy()
    m = blob._is_float_word_mask(b) & np.isnan(f)
    c[m] = 0x7FC00000
    return zlib.crc32(c.tobytes()) & 0xFFFFFFFF


def replay_golden(path, make_env):
    """Replays one tests/golden/*.npz on `make_env(**cfg)`; returns (ok, message)."""
    import json
    z = np.load(path, allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    env = make_env(**cfg)
    seed = int(z["seed"])
    if seed >= 0:
        env.seed(seed)
    env.reset(True)
    env.load(z["blob0"])
    d = blob.diff(z["blob0"], env.dump())
    if d:
        return False, "initial state does not round-trip: " + d
    acts = z["actions"]
    cps = set(int(c) for c in z["checkpoints"])
    for t in range(acts.shape[0]):
        env.take_actions(acts[t, :, :2], acts[t, :, 2].astype(np.int32))
        r = env.step()
        b = env.dump()
        if t in cps:
            d = blob.diff(z["blob_%d" % t], b)
            if d:
                return False, "step %d: %s" % (t, d)
        if golden_crc(b) != int(z["crcs"][t]):
            return False, "step %d: state CRC differs from the reference's" % t
        if not np.array_equal(np.asarray(r, np.float64), z["rewards"][t]):
            return False, "step %d: rewards %s vs %s" % (t, r, z["rewards"][t])
        if not np.array_equal(np.asarray(env.dones(), bool), z["dones"][t].astype(bool)):
            return False, "step %d: dones differ" % t
    return True, "ok"


def golden_files(gpu_capable_only=False):
    import glob, json, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    out = []
    for p in sorted(glob.glob(os.path.join(here, "*.npz"))):
        if os.path.basename(p).startswith("snapshot_"):   # JSON snapshot vectors: tests/snapshot_cases.py
            continue
        if gpu_capable_only:
            cfg = json.loads(str(np.load(p)["cfg"]))
            if cfg.get("num_agents", 1) + cfg.get("num_bots", 0) + cfg.get("example_bots", 0) > 32:    # agar_types.h AG_MAX_PLAYERS
                continue
        out.append(p)
    return out


def run_quiet_rollout(eng, oras, steps, seeds, rng_seed, check_every=100):
    """BASELINE C2's policy (action none, a fresh random direction every step) in batched lock-step: per-step
    rewards and periodic full-state blobs must be identical.  Returns (ok, message or gain count)."""
    A = len(oras)
    seeds = np.asarray(seeds, dtype=np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for o, sd in zip(oras, seeds):
        o.seed(int(sd)); o.reset(True)
    rng = np.random.RandomState(rng_seed)
    act = np.zeros((A, 1), np.int32)
    gains = 0
    for t in range(steps):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32)
        eng.set_actions(dxdy, act); eng.step()
        r = eng.rewards()
        gains += int((r > 0).sum())
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a])
            ro = oras[a].step()
            if r[a, 0] != ro[0]:
                return False, "step %d arena %d: reward %r vs %r" % (t, a, r[a, 0], ro[0])
        if t % check_every == check_every - 1 or t == steps - 1:
            for a in range(A):
                d = blob.diff(oras[a].dump(), eng.dump(a))
                if d:
                    return False, "step %d arena %d: %s" % (t, a, d)
    return True, gains   # on success: the number of (arena, step) pairs with a mass gain (evidence that pellets were eaten)


def run_engine_level_lockstep(engine, oracles, ticks, seeds, policy_seed=7, every=10, respawn=True):
    """The bench/main.cpp path in batched lock-step: Engine::tick driven directly (agarcl_tick /
    ora_tick with the envs' own dt, e.g. 1/60 s), no BaseEnvironment::step around it.  Per tick: dead players are
    respawned (the mode-0 rule), the RL agents get target = centroid + 10 * U(-1,1)^2 and a random action, bots choose
    for themselves.  The agents' absolute targets are taken from the oracle's take_action and handed to the engine
    through agarcl_set_targets; every other player's words are the engine's own.  Blobs are compared every `every` ticks."""
    A = len(oracles)
    engine.seed(np.asarray(seeds, dtype=np.uint32)); engine.reset(reset_ids=True)
    for o, s in zip(oracles, seeds):
        o.seed(int(s)); o.reset(True)
    P, n = engine.players, engine.num_agents
    for t in range(ticks):
        if respawn:
            engine.respawn_dead()
            for o in oracles:
                o.respawn_dead()
        txy = np.zeros((A, P, 2), np.float32); act = np.zeros((A, P), np.int32)
        for a, o in enumerate(oracles):
            dxdy, ac = policy(policy_seed + 7919 * a, t, n, True, 1)
            for i, pid in enumerate(o.pids()):
                o.take_action(pid, float(dxdy[i, 0]), float(dxdy[i, 1]), int(ac[i]))
            want = {p["pid"]: p for p in blob.parse(o.dump())["players"]}
            _, pl = engine.arena_words(a)
            for slot in range(P):
                pid, kind = int(pl[slot, 15]), int(pl[slot, 16])      # PL_PID, PL_KIND (agar_types.h)
                if kind == 0:                                         # RL agent: the oracle's take_action result
                    txy[a, slot] = want[pid]["target"]; act[a, slot] = want[pid]["action"]
                else:                                                 # bot: whatever the engine's own bot logic chose
                    txy[a, slot] = pl[slot, 2:4].view(np.float32); act[a, slot] = pl[slot, 1]
        engine.set_targets(txy, act)
        engine.tick(1)
        for o in oracles:
            o.tick()
        if (t + 1) % every == 0 or t == ticks - 1:
            fl = engine.flags()
            if fl.any():
                return False, "tick %d: capacity flags raised %s" % (t, fl.tolist())
            for a in range(A):
                d = blob.diff(oracles[a].dump(), engine.dump(a))
                if d:
                    return False, "tick %d: arena %d: %s" % (t, a, d)
    return True, "ok"


def soak_trial(seed, trial):
    """The draws scripts/gpu_soak.py makes (default distribution) up to and including `trial` of stream `seed`:
    (cfg, launch pins, arenas, arena seeds, policy seed, sticky) of that trial -- so that a recorded soak finding can be replayed."""
    rng = np.random.RandomState(seed)
    for _ in range(trial + 1):
        na = int(rng.choice([1, 1, 1, 1, 2, 3])); mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 6, 7, 8, 9, 10]))
        nb = int(rng.randint(0, 4)) if mode == 0 and rng.rand() < 0.4 else 0
        if mode > 6: na = 1
        cfg = dict(num_agents=na, arena_size=int(rng.choice([80, 150, 250, 400, 1000, 1100])), num_pellets=int(rng.choice([50, 64, 200, 500, 1000, 1300])),
                   num_viruses=int(rng.choice([0, 0, 3, 10, 25])), num_bots=nb, mode=mode, reward_type=int(rng.randint(0, 2)), c_death=int(rng.choice([0, -20])))
        pins = dict(AGARCL_TILE_LG=str(rng.choice([0, 6])), AGARCL_FUSED=str(rng.choice([0, 1])), AGARCL_FUSED_QG=str(rng.choice([1, 2, 4, 8, 16, 16])), AGARCL_QUIET_QG=str(rng.choice([1, 2, 4, 8, 16])))
        A = int(rng.choice([3, 70, 130]))
        if mode in (1, 2, 5) and (cfg["arena_size"] // 2) * 4 > 2048: continue   # squared pellets beyond the pellet capacity: agarcl_create rejects, the soak skips
        sd, ps, st = rng.randint(1, 1 << 30, size=A), int(rng.randint(1, 1000)), int(rng.choice([1, 4, 8]))
    return cfg, pins, A, sd, ps, st
