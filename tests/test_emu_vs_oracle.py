"""Kernel LOGIC on the CPU: the wave-level source of the HIP kernel compiled with lanes-as-loops
(tests/emu, test-only) against the oracle in batched lock-step.  The GPU tests repeat this on hardware."""
import numpy as np
import pytest
from lockstep import run_batched_lockstep, run_engine_level_lockstep, run_quiet_rollout

CASES = [
    (dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0), 300, 4),
    (dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6), 1200, 16),
    (dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6), 1000, 8),
    (dict(arena_size=300, num_pellets=300, num_viruses=5, mode=5), 500, 8),
    (dict(arena_size=300, num_pellets=300, num_viruses=5, mode=1), 200, 8),
    (dict(arena_size=1200, num_pellets=800, num_viruses=15, mode=3), 200, 8),
    (dict(arena_size=60, num_pellets=200, num_viruses=0, mode=0), 200, 8),
    (dict(arena_size=50, num_pellets=200, num_viruses=0, mode=0), 300, 8),      # the gym "trivial" preset (AgarioEnv.py:329-338) ...
    (dict(arena_size=50, num_pellets=200, num_viruses=0, mode=6), 300, 8),      # ... and with a mass-1000 agent: a sixth of the arena under one cell
    # the soak's capacity corner: 80 x 80 with 1300 pellets -- a mass-1000 cell has ~200 pellets within its radius, a few such cells
    # ate more than the 256 events per tick the fixed arrays held (flag 0x8): dense arenas spill the further events to HBM now (16 861 in one tick here)
    (dict(arena_size=80, num_pellets=1300, num_viruses=0, mode=6), 150, 8),
    (dict(num_agents=2, arena_size=80, num_pellets=1300, num_viruses=0, num_bots=2, mode=0), 150, 8),
    # several players per arena: bots (mode 0, modes 7-10) and multi-agent (cell-eats-cell, map order)
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0), 600, 4),
    (dict(num_agents=3, arena_size=250, num_pellets=500, num_viruses=10, mode=6), 500, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=7), 300, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=8), 300, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=9), 300, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=10), 300, 8),
    (dict(num_agents=14, arena_size=300, num_pellets=300, num_viruses=5, mode=0), 60, 8),   # 14th insert rehashes the player map
    (dict(num_agents=2, arena_size=150, num_pellets=300, num_viruses=3, num_bots=3, mode=0, reward_type=0), 400, 8),
    # ExampleBots (bench/main.cpp's population): 31 players in one arena, and the four bot kinds beside them
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, mode=6, example_bots=30), 200, 8),
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, example_bots=12), 300, 4),
    (dict(num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=0), 100, 4),      # Tick/0: an engine without players
    (dict(num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=5), 100, 4),      # Tick/5
    # crowded: players eat each other all the time; bot ticks with and without scripted bots; the remembered pellet verdicts across a regeneration (tick 120)
    (dict(num_agents=3, arena_size=120, num_pellets=300, num_viruses=4, num_bots=6, mode=0), 200, 4),
    (dict(num_agents=4, arena_size=250, num_pellets=500, num_viruses=10, mode=6, example_bots=20), 120, 4),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_emulated_kernel_logic_matches_oracle(emu_lib, oracle_lib, case):
    from agarcl_amd import _capi
    cfg, steps, sticky = CASES[case]
    A = 4
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, steps, seeds=np.arange(11, 11 + A), sticky=sticky, every=2)
    assert ok, "%s: %s" % (cfg, msg)


@pytest.mark.parametrize("case", [0, 1, 7, 8])
def test_emulated_kernel_logic_tile_transposed_layout(emu_lib, oracle_lib, monkeypatch, case):
    """The per-arena word arrays in tiles of 64 arenas (agar_types.h; chosen for big single-player batches, AGARCL_TILE_LG=6 pins
    it): same results.  70 arenas = one full tile and a ragged one; quiet, full-ruleset, bots and multi-agent configs."""
    from agarcl_amd import _capi
    monkeypatch.setenv("AGARCL_TILE_LG", "6")
    cfg, steps, sticky = CASES[case]
    A = 70
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, min(steps, 120), seeds=np.arange(411, 411 + A), sticky=sticky, every=10)
    assert ok, "%s: %s" % (cfg, msg)


@pytest.mark.parametrize("cfg", [
    dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0),     # BASELINE C2
    dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=0),    # C3 / mode 0
    dict(arena_size=1400, num_pellets=1500, num_viruses=0, mode=0),     # 3x3 pellet grid: bucket visibility matters (no AV)
    dict(arena_size=300, num_pellets=400, num_viruses=0, mode=1),       # squared pellets, no regen, no decay
    dict(arena_size=100, num_pellets=60, num_viruses=0, mode=0),        # crowded: two pellets eaten in one tick (inline), regeneration
    dict(arena_size=60, num_pellets=200, num_viruses=0, mode=0),        # very crowded: double eats in either scan order, triple eats handed over
    dict(arena_size=1100, num_pellets=2000, num_viruses=0, mode=3),     # 3x3 pellet grid (bucket ranks decide the eat order), dense
])
def test_emulated_quiet_path_long_rollout(emu_lib, oracle_lib, cfg):
    """The front kernel's logic (agar_quiet.inl + quiet_ticks: pellet-free disc, inline eat / decay, hand-over to the
    general path) under the C2 policy for 1500 steps."""
    from agarcl_amd import _capi
    A = 3
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_quiet_rollout(eng, oras, 1500, 31000 + np.arange(A), rng_seed=17)
    assert ok, "%s: %s" % (cfg, msg)
    assert msg > 0    # somebody ate


def test_emulated_quiet_run_with_lazily_loaded_pellets(emu_lib, oracle_lib, monkeypatch):
    """AGARCL_NO_FRONT=1: every arena-step goes through the general engine's own quiet run, whose launch has NOT read the pellets when the
    tracked pellet is eaten without a pass -- RegPel::swap_pop must fetch them first (without it the swap wrote unloaded
    registers back and a later load undid it: 'step 99 arena 2: pellet_x[520] 274.2 vs 0.0')."""
    from agarcl_amd import _capi
    monkeypatch.setenv("AGARCL_NO_FRONT", "1")
    cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
    A = 8
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_quiet_rollout(eng, oras, 400, 31000 + np.arange(A), rng_seed=17)
    assert ok, msg
    assert msg > 0    # somebody ate


def test_emulated_engine_level_60hz(emu_lib, oracle_lib):
    """BASELINE configs[0] as bench/main.cpp drives it: Engine::tick at dt = 1/60 s (600-tick recombine deadlines), agent +
    the four bot kinds on the default 250x250 arena, through agarcl_set_targets / agarcl_tick on the kernel source."""
    from agarcl_amd import _capi
    cfg = dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
    A = 2
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_engine_level_lockstep(eng, oras, 1500, seeds=[42, 43], every=25)
    assert ok, msg
    # a single mass-1000 agent: splits at tick ~0, so recombination needs the full 600-tick deadline
    cfg = dict(num_agents=1, arena_size=300, num_pellets=300, num_viruses=5, mode=6, dt=1.0 / 60)
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_engine_level_lockstep(eng, oras, 1400, seeds=[5, 6], every=25, respawn=False)
    assert ok, msg


def test_emulated_mode3_done_threshold(emu_lib, oracle_lib):
    """Mode 3 ends an episode when the agent's mass reaches 23 000 (BaseEnvironment.hpp:108-111; pinned against the reference in
    test_oracle_vs_reference.py): cells loaded just below the threshold cross it by eating -- through the lean front part (arena 0:
    sparse pellets) and through the general engine (crowded arenas) -- dones, rewards and state against the oracle."""
    from oracle import blob
    from agarcl_amd import _capi
    cfg = dict(num_agents=1, arena_size=300, num_pellets=600, num_viruses=0, mode=3, reward_type=1)
    A = 4
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    seeds = np.array([9, 10, 11, 12], dtype=np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for a, o in enumerate(oras):
        o.seed(int(seeds[a])); o.reset(True)
        d = blob.parse(o.dump()); d["players"][0]["cell_mass"][0] = 22999 - 3 * a
        b = blob.build(d); o.load(b); eng.load(b, a)
    rng = np.random.RandomState(4)
    seen = np.zeros(A, bool)
    for t in range(100):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = np.zeros((A, 1), np.int32)
        eng.set_actions(dxdy, act); eng.step()
        r, dn = eng.rewards(), eng.dones()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); ro = oras[a].step()
            assert r[a, 0] == ro[0], (t, a)
            assert bool(dn[a, 0]) == bool(np.asarray(oras[a].dones())[0]), (t, a)
        seen |= dn[:, 0].astype(bool)
    for a in range(A):
        assert blob.diff(oras[a].dump(), eng.dump(a)) is None, a
    assert seen.all()
    eng.close()


def test_emulated_negative_decay_factor(emu_lib, oracle_lib):
    """Entities.hpp:199-202 in the reference's x86-64 build: with >= 66 virus meals inside the anti-team window the decay factor
    1 - 0.002 * 1.1^k is negative and the double -> uint32 conversion wraps (mass 2^32 - x, not 0).  A player loaded with 70 recent
    virus meals reaches its next decay check: the same wrapped mass as the reference (state equal right after that step), and -- the
    mass now exceeds the radius / speed tables -- the capacity flag instead of a silent divergence."""
    from oracle import blob
    from agarcl_amd import _capi
    cfg = dict(num_agents=1, arena_size=300, num_pellets=100, num_viruses=0, mode=0, reward_type=1)
    A = 3
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    seeds = np.array([3, 4, 5], dtype=np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for a, o in enumerate(oras):
        o.seed(int(seeds[a])); o.reset(True)
        d = blob.parse(o.dump())
        p = d["players"][0]; p["cell_mass"][0] = 3000 + 100 * a; p["elapsed"] = 500; p["last_decay"] = 470 - a; p["anti_team"] = np.float32(1.1 ** 69)
        p["virus_ticks"] = np.arange(400, 470, dtype=np.int64)
        b = blob.build(d); o.load(b); eng.load(b, a)
    rng = np.random.RandomState(2)
    wrapped = np.zeros(A, bool)
    for t in range(40):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = np.zeros((A, 1), np.int32)
        eng.set_actions(dxdy, act); eng.step()
        r, fl = eng.rewards(), eng.flags()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); ro = oras[a].step()
            if wrapped[a]:
                assert fl[a] & 32      # flagged from the wrap on: a mass beyond the 2^19-entry tables has left the reference
                continue
            assert r[a, 0] == ro[0], (t, a)
            assert blob.diff(oras[a].dump(), eng.dump(a)) is None, (t, a)      # (the decay is the step's last tick: still equal, wrapped mass included)
            wrapped[a] = int(blob.parse(oras[a].dump())["players"][0]["cell_mass"][0]) > (1 << 31)
            assert bool(fl[a] & 32) == bool(wrapped[a]), "the flag goes up with the wrapped mass, in the same step, and not before"
    assert (eng.flags() & 32).all() and wrapped.all()      # every arena got there: same wrapped mass as the reference, and the flag
    eng.close()


@pytest.mark.timeout(120)
def test_emulated_wrapped_mass_terminates(emu_lib):
    """A mass that has wrapped around 2^32 (the reference's unsigned arithmetic allows it: negative decay factor) must not hang the
    growth closure of the pellet scan (m + K wraps back to a tiny mass and the candidate count would oscillate): the step ends
    and the arena carries AGARCL_F_MASS_LUT_OVERFLOW."""
    from oracle import blob
    from agarcl_amd import _capi
    cfg = dict(num_agents=1, arena_size=80, num_pellets=64, num_viruses=0, mode=0)
    A = 2
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    eng.seed(np.array([5, 6], dtype=np.uint32)); eng.reset(reset_ids=True)
    for a in range(A):
        d = blob.parse(eng.dump(a)); d["players"][0]["cell_mass"][0] = (1 << 32) - 3 - a
        eng.load(blob.build(d), a)
    for t in range(8):
        eng.set_actions(np.zeros((A, 1, 2), np.float32), np.zeros((A, 1), np.int32)); eng.step()
    assert (eng.flags() & 32).all()
    eng.close()


def test_emulated_mass_beyond_tables_is_flagged(emu_lib):
    """The reference computes radius and speed from the mass without a limit; the tables end at 2^19.  A cell beyond that makes every
    look-up clamp, so the arena must carry AGARCL_F_MASS_LUT_OVERFLOW after the very next step -- through the lean front part (arena 0:
    one quiet cell) and through the general engine (arena 1: viruses around, several cells after a split)."""
    from oracle import blob
    from agarcl_amd import _capi
    cfg = dict(num_agents=1, arena_size=400, num_pellets=200, num_viruses=5, mode=0)
    A = 3
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    eng.seed(np.array([5, 6, 7], dtype=np.uint32)); eng.reset(reset_ids=True)
    for a in (0, 1):
        d = blob.parse(eng.dump(a)); d["players"][0]["cell_mass"][0] = 600000 + a
        if a == 0: d["virus_x"] = d["virus_x"][:0]; d["virus_y"] = d["virus_y"][:0]; d["virus_vx"] = d["virus_vx"][:0]; d["virus_vy"] = d["virus_vy"][:0]; d["virus_mass"] = d["virus_mass"][:0]; d["virus_hits"] = d["virus_hits"][:0]; d["virus_id"] = d["virus_id"][:0]
        eng.load(blob.build(d), a)
    eng.set_actions(np.zeros((A, 1, 2), np.float32), np.array([[0], [2], [0]], np.int32)); eng.step()
    fl = eng.flags()
    assert fl[0] & 32 and fl[1] & 32 and fl[2] == 0, fl
    eng.close()


def test_batched_simple_turns_keep_the_iteration_order_where_it_matters(emu_lib, oracle_lib):
    """several players per arena: the turns that are pure bookkeeping are performed for all such players at once (agar_core.inl simple_turns),
    out of the engine's iteration order -- except on ticks on which a player ejects food, which the players behind it must test.  16 crowded
    arenas, agents that feed and split at random: the first form of the batching, without that exception, diverged here at step 94."""
    from agarcl_amd import _capi
    cfg = dict(num_agents=2, arena_size=80, num_pellets=1300, num_viruses=0, num_bots=2, mode=0)
    A = 16
    eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 130, seeds=np.arange(500, 500 + A), sticky=8, every=5)
    assert ok, msg
