"""Pins the C restatement (oracle/agar_oracle.c) against the REAL reference engine compiled from
/root/reference (oracle/_ref/libagar_ref.so): same seeds, same actions, every tick, word for word."""
import pytest
from lockstep import run_lockstep

CASES = [
    # (env kwargs, ticks, sticky)  -- SURVEY.md 8(d) configs C1..C3 + modes / bots / multi-agent
    (dict(num_agents=1, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0), 1500, 4),
    (dict(num_agents=1, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0), 1500, 4),
    (dict(num_agents=1, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6), 3000, 16),
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0), 3000, 4),
    (dict(num_agents=3, arena_size=250, num_pellets=500, num_viruses=10, mode=6), 2500, 8),
    (dict(num_agents=1, arena_size=300, num_pellets=300, num_viruses=5, mode=5), 1500, 8),
    (dict(num_agents=1, arena_size=300, num_pellets=300, num_viruses=5, mode=1), 800, 8),
    (dict(num_agents=1, arena_size=300, num_pellets=300, num_viruses=5, mode=2), 800, 8),
    (dict(num_agents=1, arena_size=1200, num_pellets=600, num_viruses=12, mode=3), 800, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=7), 1500, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=8), 1500, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=9), 1500, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=10), 1500, 8),
    (dict(num_agents=14, arena_size=300, num_pellets=300, num_viruses=5, mode=0), 300, 8),  # 14th insert rehashes the player map
    # bench/main.cpp's population: the reference's ExampleBot (agario/bots/ExampleBot.hpp) -- alone, and as prey of agents (29 -> 59 rehash at 30+)
    (dict(num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=30), 1500, 4),
    (dict(num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=0), 600, 4),     # Tick/0: nobody at all
    (dict(num_agents=2, arena_size=250, num_pellets=500, num_viruses=10, mode=6, example_bots=20), 1500, 8),
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, example_bots=10, dt=1.0 / 60), 1500, 4),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("seed", [1, 42])
def test_lockstep_bit_exact(ref_lib, oracle_lib, case, seed):
    cfg, ticks, sticky = CASES[case]
    r = ref_lib.RefEnv(**cfg)
    o = oracle_lib.OraEnv(**cfg)
    ok, msg = run_lockstep([r, o], ticks, seed, policy_seed=seed + 5, sticky=sticky)
    assert ok, "%s seed %d: %s" % (cfg, seed, msg)


def test_bench_dt_60hz(ref_lib, oracle_lib):
    """bench/main.cpp path: dt = 1/60 (600-tick recombine timer)."""
    cfg = dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
    r = ref_lib.RefEnv(**cfg); o = oracle_lib.OraEnv(**cfg)
    ok, msg = run_lockstep([r, o], 3000, 42, policy_seed=7, sticky=4)
    assert ok, msg


def test_env_step_rewards_dones(ref_lib, oracle_lib):
    """BaseEnvironment::step rewards / dones (BaseEnvironment.hpp:89-122) incl. mode-3 done and reward_type 0."""
    import numpy as np
    from lockstep import policy
    from oracle import blob
    for cfg in (dict(num_agents=1, arena_size=300, num_pellets=400, num_viruses=4, mode=0, reward_type=1),
                dict(num_agents=2, arena_size=300, num_pellets=400, num_viruses=4, mode=6, reward_type=0),
                dict(num_agents=1, arena_size=300, num_pellets=400, num_viruses=4, mode=3, reward_type=1),
                dict(num_agents=1, arena_size=200, num_pellets=300, num_viruses=3, num_bots=1, mode=9, reward_type=1)):
        r = ref_lib.RefEnv(**cfg); o = oracle_lib.OraEnv(**cfg)
        for e in (r, o):
            e.seed(5); e.reset(True)
        for t in range(400):
            dxdy, act = policy(3, t, cfg["num_agents"], True, 8)
            r.take_actions(dxdy, act); o.take_actions(dxdy, act)
            assert np.array_equal(r.step(), o.step()), (cfg, t)
            assert np.array_equal(r.dones(), o.dones()), (cfg, t)
            if t % 20 == 0:
                assert blob.diff(r.dump(), o.dump()) is None


def test_screen_respawn_hook_pinned(ref_lib, oracle_lib):
    """E5: the oracle's ScreenEnvironment hook path against the REAL BaseEnvironment::step with the hook's three statements
    (ScreenEnvironment.hpp:233-243) plugged into its _partial_observation override: rewards (c_death term), dones, full state."""
    import numpy as np
    from lockstep import policy
    from oracle import blob
    for cfg in (dict(num_agents=2, arena_size=120, num_pellets=150, num_viruses=2, num_bots=0, mode=4, c_death=-50),
                dict(num_agents=1, arena_size=150, num_pellets=200, num_viruses=3, num_bots=3, mode=0, c_death=-7),
                dict(num_agents=1, arena_size=150, num_pellets=200, num_viruses=3, num_bots=1, mode=8, c_death=-3)):
        r = ref_lib.RefEnv(**cfg); o = oracle_lib.OraEnv(**cfg)
        r.set_screen_hook(True); o.set_screen_hook(True)
        for e in (r, o):
            e.seed(9); e.reset(True)
        respawns = 0
        for t in range(700):
            dxdy, act = policy(4, t, cfg["num_agents"], True, 6)
            r.take_actions(dxdy, act); o.take_actions(dxdy, act)
            rr, ro = r.step(), o.step()
            assert np.array_equal(rr, ro), (cfg, t, rr, ro)
            assert np.array_equal(r.dones(), o.dones()), (cfg, t)
            if t % 10 == 0:
                assert blob.diff(r.dump(), o.dump()) is None, (cfg, t)
        assert blob.diff(r.dump(), o.dump()) is None


def test_mode3_done_threshold_pinned(ref_lib, oracle_lib):
    """Mode 3 ends an episode when the agent's mass reaches 23 000 (BaseEnvironment.hpp:108-111): a cell loaded at mass 22 990
    grows across the threshold by eating pellets; rewards, dones and state of the oracle against the real reference."""
    import numpy as np
    from oracle import blob
    cfg = dict(num_agents=1, arena_size=300, num_pellets=600, num_viruses=0, mode=3, reward_type=1)
    r = ref_lib.RefEnv(**cfg); o = oracle_lib.OraEnv(**cfg)
    for e in (r, o):
        e.seed(9); e.reset(True)
    d = blob.parse(o.dump()); d["players"][0]["cell_mass"][0] = 22990; b = blob.build(d)
    r.load(b); o.load(b)
    rng = np.random.RandomState(4)
    seen_done = False
    for t in range(120):
        dxdy = rng.uniform(-1, 1, size=(1, 2)).astype(np.float32); act = np.zeros(1, np.int32)
        r.take_actions(dxdy, act); o.take_actions(dxdy, act)
        assert np.array_equal(r.step(), o.step()), t
        assert np.array_equal(r.dones(), o.dones()), t
        seen_done = seen_done or bool(np.asarray(r.dones()).any())
    assert blob.diff(r.dump(), o.dump()) is None
    assert seen_done, "the threshold was never crossed: the test does not test what it says"


def test_negative_decay_factor_wraps_pinned(ref_lib, oracle_lib):
    """Entities.hpp:199-202 in the reference's x86-64 build: with >= 66 virus meals inside the anti-team window the decay factor
    1 - 0.002 * 1.1^k is negative and the double -> uint32 conversion wraps (mass 2^32 - x, not 0).  A player loaded with 70 recent
    virus meals reaches its next decay check: the wrapped mass, and everything that follows from it, like the reference."""
    from oracle import blob
    cfg = dict(num_agents=1, arena_size=300, num_pellets=100, num_viruses=0, mode=0, reward_type=1)
    import numpy as np
    r = ref_lib.RefEnv(**cfg); o = oracle_lib.OraEnv(**cfg)
    for e in (r, o):
        e.seed(3); e.reset(True)
    d = blob.parse(o.dump())
    p = d["players"][0]; p["cell_mass"][0] = 3000; p["elapsed"] = 500; p["last_decay"] = 470; p["anti_team"] = np.float32(1.1 ** 69)
    p["virus_ticks"] = np.arange(400, 470, dtype=np.int64)
    b = blob.build(d); r.load(b); o.load(b)
    rng = np.random.RandomState(2)
    wrapped = False
    for t in range(40):
        dxdy = rng.uniform(-1, 1, size=(1, 2)).astype(np.float32); act = np.zeros(1, np.int32)
        r.take_actions(dxdy, act); o.take_actions(dxdy, act)
        assert np.array_equal(r.step(), o.step()), t
        dd = blob.parse(r.dump())
        wrapped = wrapped or int(dd["players"][0]["cell_mass"][0]) > (1 << 31)
        assert blob.diff(r.dump(), o.dump()) is None, t
    assert wrapped, "the decay never produced a wrapped mass: the test does not test what it says"
