"""Parity tests proper: the HIP engine (through the C ABI, on a real MI355X) against the C oracle on the
same seeded inputs, against the golden vectors recorded from the real reference, and -- at BASELINE.json's
full size (4096 arenas) -- through size-independent properties.  Bar: bit-exact (ints AND fp32 words)."""
import os

import numpy as np
import pytest

from lockstep import EngineAsEnv, golden_files, replay_golden, run_batched_lockstep, run_engine_level_lockstep, run_quiet_rollout, policy

pytestmark = pytest.mark.gpu

C2 = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
C3 = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
C3M6 = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)


@pytest.mark.parametrize("path", golden_files(gpu_capable_only=True), ids=lambda p: os.path.basename(p)[:-4])
def test_hip_matches_reference_golden(hip_engine_cls, path):
    ok, msg = replay_golden(path, lambda **cfg: EngineAsEnv(hip_engine_cls, **cfg))
    assert ok, msg


@pytest.mark.parametrize("cfg,steps,sticky", [
    (C2, 400, 4), (C3, 400, 4), (C3M6, 1000, 16), (C3M6, 600, 3),
    (dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6), 1000, 8),
    (dict(arena_size=300, num_pellets=300, num_viruses=5, mode=5), 600, 8),
    (dict(arena_size=300, num_pellets=300, num_viruses=5, mode=1), 300, 8),
    (dict(arena_size=300, num_pellets=300, num_viruses=5, mode=2), 300, 8),
    (dict(arena_size=1200, num_pellets=800, num_viruses=15, mode=3), 300, 8),
    (dict(arena_size=60, num_pellets=200, num_viruses=0, mode=0), 300, 8),   # "trivial" difficulty-like tiny arena
    # the gym "trivial" preset itself (AgarioEnv.py:329-338: 50 x 50, 200 pellets): next to the capacity corner of the dense small arenas
    (dict(arena_size=50, num_pellets=200, num_viruses=0, mode=0), 400, 8),
    (dict(arena_size=50, num_pellets=200, num_viruses=0, mode=6), 400, 8),
    # the capacity corner of the dense small arenas: more than 256 eat events in one tick: candidates up to the pellet capacity, further events spilled to HBM
    (dict(arena_size=80, num_pellets=1300, num_viruses=0, mode=6), 300, 8),
    (dict(num_agents=2, arena_size=80, num_pellets=1300, num_viruses=0, num_bots=2, mode=0), 300, 8),    # four players: 37 821 eat events in one tick
    (dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6, reward_type=0), 200, 8),
    # several players per arena: bots, multi-agent, map-order rehash
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0), 600, 4),
    (dict(num_agents=3, arena_size=250, num_pellets=500, num_viruses=10, mode=6), 500, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=7), 300, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=8), 300, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=9), 300, 8),
    (dict(num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=10), 300, 8),
    (dict(num_agents=14, arena_size=300, num_pellets=300, num_viruses=5, mode=0), 60, 8),
    (dict(num_agents=2, arena_size=150, num_pellets=300, num_viruses=3, num_bots=3, mode=0, reward_type=0), 400, 8),
    # bench/main.cpp's ExampleBots: 21 and 31 players per arena (players-map rehashes 13 -> 29 -> 59), and beside the four scripted kinds
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=20), 300, 4),
    (dict(num_agents=1, arena_size=300, num_pellets=500, num_viruses=10, mode=6, example_bots=30), 200, 8),
    (dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, example_bots=12), 300, 4),
    (dict(num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=0), 100, 4),      # Tick/0: an engine without players
    (dict(num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, example_bots=30), 150, 4),     # Tick/30
    # crowded arenas in which players eat each other all the time: players_collision's lane-parallel strip scan and the eats applied from its records
    # (r05; the oracle replays the reference's sequential solve()), the scripted bots' lane-parallel checks, bot ticks with and without them
    (dict(num_agents=3, arena_size=120, num_pellets=300, num_viruses=4, num_bots=6, mode=0), 400, 4),
    (dict(num_agents=2, arena_size=200, num_pellets=400, num_viruses=5, num_bots=12, mode=0, example_bots=8), 300, 4),
    (dict(num_agents=4, arena_size=250, num_pellets=500, num_viruses=10, mode=6, example_bots=20), 200, 4),
])
def test_hip_vs_oracle_lockstep(hip_engine_cls, oracle_lib, cfg, steps, sticky):
    A = 16
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, steps, seeds=np.arange(500, 500 + A), sticky=sticky, every=5)
    eng.close()
    assert ok, "%s: %s" % (cfg, msg)


@pytest.mark.parametrize("pins", [dict(AGARCL_KSTEP_GRID="5"), dict(AGARCL_KSTEP_GRID="3", AGARCL_NO_ORDER="1"), dict(AGARCL_KSTEP_GRID="7", AGARCL_TILE_LG="6")],
                         ids=["grid5-order", "grid3-index", "grid7-tiled"])
@pytest.mark.parametrize("cfg", [C3M6, dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0)], ids=["mode6", "c1"])
def test_several_arenas_per_workgroup(hip_engine_cls, oracle_lib, monkeypatch, cfg, pins):
    """k_step with fewer workgroups than arenas (AGARCL_KSTEP_GRID caps the grid: what batches beyond 4096 arenas do): arenas beyond the grid are
    drawn from the work counter, by descending cost of their last visit once k_order has run (every 8th step), or in index order -- every arena
    against the oracle, 60 steps."""
    for k, v in pins.items():
        monkeypatch.setenv(k, v)
    A = 29
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 60, seeds=np.arange(900, 900 + A), sticky=8, every=3)
    eng.close()
    assert ok, "%s %s: %s" % (cfg, pins, msg)


def test_every_arena_equal_across_launch_forms_8192(hip_engine_cls, oracle_lib, monkeypatch):
    """8192 arenas of the full rule set (two items per workgroup): the per-arena digest (counts, masses, rewards, flags) after 16 steps is the
    same for the default launch, for index order, for a small grid and for the tiled layout -- EVERY arena, not a sample (a real-call form of
    the quiet run once passed six sampled arenas while ~3000 of 8192 differed: DESIGN.md section 5) -- and 48 sampled arenas equal the oracle."""
    from oracle import blob
    A, steps = 8192, 16
    rng = np.random.RandomState(4)
    acts = [(rng.uniform(-1, 1, (A, 1, 2)).astype(np.float32), rng.randint(0, 3, (A, 1)).astype(np.int32)) for _ in range(steps)]

    def run(pins, sample=()):
        for k in ("AGARCL_KSTEP_GRID", "AGARCL_NO_ORDER", "AGARCL_TILE_LG"):
            monkeypatch.delenv(k, raising=False)
        for k, v in pins.items():
            monkeypatch.setenv(k, v)
        eng = hip_engine_cls(A, **C3M6)
        eng.seed(None, 52000); eng.reset(reset_ids=True)
        for d, a in acts:
            eng.set_actions(d, a); eng.step()
        dig = np.concatenate([eng.counts(), eng.masses().reshape(A, -1), eng.rewards().reshape(A, -1).astype(np.int64), eng.flags().reshape(A, 1).astype(np.int64)], axis=1)
        dumps = {a: eng.dump(a) for a in sample}
        eng.close()
        return dig, dumps
    sample = [int(x) for x in np.random.RandomState(1).choice(A, 48, replace=False)]
    ref, dumps = run({}, sample)
    assert not ref[:, -1].any()
    for pins in (dict(AGARCL_NO_ORDER="1"), dict(AGARCL_KSTEP_GRID="1024"), dict(AGARCL_TILE_LG="6")):
        got, _ = run(pins)
        bad = np.nonzero((got != ref).any(axis=1))[0]
        assert len(bad) == 0, "%s: %d arenas differ from the default launch, first %s" % (pins, len(bad), bad[:8])
    for a in sample:
        o = oracle_lib.OraEnv(**C3M6); o.seed(52000 + a); o.reset(True)
        for d, ac in acts:
            o.take_actions(d[a], ac[a]); o.step()
        assert blob.diff(o.dump(), dumps[a]) is None, "arena %d" % a


@pytest.mark.parametrize("A", [1, 5, 7, 67])
def test_front_kernel_odd_arena_counts_long_quiet_rollout(hip_engine_cls, oracle_lib, A):
    """The lean front kernel (agar_quiet.inl) packs 4 arenas per wavefront: arena counts that are not a multiple of
    4 exercise its padding groups; 1200 steps of the C2 policy (action none, fresh direction every step) exercise
    the pellet-free-disc rule, inline eats, decay and the hand-over to k_step on regen ticks -- all bit-exact."""
    eng = hip_engine_cls(A, **C2)
    oras = [oracle_lib.OraEnv(**C2) for _ in range(A)]
    ok, msg = run_quiet_rollout(eng, oras, 1200, 20000 + np.arange(A), rng_seed=A)
    assert ok, msg
    assert msg > 0      # somebody ate
    eng.close()


@pytest.mark.parametrize("cfg", [
    dict(arena_size=1400, num_pellets=1500, num_viruses=0, mode=0),     # 32 pellet slots per lane, 3x3 pellet grid (bucket visibility)
    dict(arena_size=1400, num_pellets=2000, num_viruses=30, mode=0),    # near the pellet capacity
    dict(arena_size=100, num_pellets=60, num_viruses=0, mode=0),        # 4 slots, crowded: frequent eats and regeneration
    dict(arena_size=500, num_pellets=450, num_viruses=3, mode=3),       # 8 slots, mode 3 (no decay)
    dict(arena_size=300, num_pellets=400, num_viruses=0, mode=1),       # squared pellets, no regeneration, no decay
])
def test_quiet_rollout_other_instantiations(hip_engine_cls, oracle_lib, cfg):
    """Every pellet-slot instantiation (4 / 8 / 16 / 32) and the bucket-visibility variant of the front part under the
    C2 policy for 800 steps, bit-exact against the oracle."""
    A = 6
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_quiet_rollout(eng, oras, 800, 7000 + np.arange(A), rng_seed=3)
    eng.close()
    assert ok, "%s: %s" % (cfg, msg)


def test_random_configurations_fuzz(hip_engine_cls, oracle_lib):
    """Differential fuzz: random single- and multi-player configurations, random actions, 150 steps each."""
    rng = np.random.RandomState(2024)
    for trial in range(14):
        na = int(rng.choice([1, 1, 1, 2, 3]))
        mode = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))
        nb = int(rng.randint(0, 5)) if mode == 0 else 0
        if mode > 6:
            na = 1
        cfg = dict(num_agents=na, arena_size=int(rng.choice([80, 150, 250, 400, 1000, 1100])), num_pellets=int(rng.choice([50, 64, 200, 500, 1000, 1300])),
                   num_viruses=int(rng.choice([0, 3, 10, 25])), num_bots=nb, mode=mode, reward_type=int(rng.randint(0, 2)), c_death=int(rng.choice([0, -20])))
        A = 4
        eng = hip_engine_cls(A, **cfg)
        oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
        ok, msg = run_batched_lockstep(eng, oras, 150, seeds=rng.randint(1, 1 << 30, size=A), policy_seed=int(rng.randint(1, 1000)), sticky=int(rng.choice([1, 4, 8])), every=10)
        eng.close()
        assert ok, "trial %d %s: %s" % (trial, cfg, msg)


def test_step_mode_adapts_and_results_do_not_depend_on_it(hip_engine_cls, oracle_lib):
    """Mode-0 arenas whose agents have become big (mass 3000 with random split / eject actions) need the general engine
    every step: the engine starts with the fused single-launch step, notices through its asynchronous statistics that
    the front part finishes almost nothing, and switches to the two-kernel step -- bit-exact results throughout."""
    import ctypes as C
    from oracle import blob
    A, steps = 64, 400
    eng = hip_engine_cls(A, **C3)
    eng.L.agarcl_debug_fused.argtypes = [C.c_void_p]
    oras = [oracle_lib.OraEnv(**C3) for _ in range(A)]
    seeds = (900 + np.arange(A)).astype(np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for a, o in enumerate(oras):
        o.seed(int(seeds[a])); o.reset(True)
        d = blob.parse(o.dump()); d["players"][0]["cell_mass"][0] = 3000
        b = blob.build(d); o.load(b); eng.load(b, a)
    assert eng.L.agarcl_debug_fused(eng.h) == 1
    rng = np.random.RandomState(1)
    for t in range(steps):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = rng.randint(0, 3, size=(A, 1)).astype(np.int32)
        eng.set_actions(dxdy, act); eng.step()
        r = eng.rewards()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); ro = oras[a].step()
            assert r[a, 0] == ro[0], (t, a)
    for a in range(A):
        assert blob.diff(oras[a].dump(), eng.dump(a)) is None, a
    assert eng.L.agarcl_debug_fused(eng.h) == 0, "the engine should have left the fused step mode"
    eng.close()


def test_front_kernel_on_off_equivalence(hip_engine_cls, monkeypatch):
    """AGARCL_NO_FRONT=1 (diagnostic switch) runs everything through k_step; results must not depend on it."""
    A, steps = 64, 300
    outs = []
    for nf in ("0", "1"):
        monkeypatch.setenv("AGARCL_NO_FRONT", nf)
        eng = hip_engine_cls(A, **C3)
        eng.seed(None, 4242); eng.reset(reset_ids=True)
        rng = np.random.RandomState(5)
        rs = []
        for t in range(steps):
            eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32))
            eng.step(); rs.append(eng.rewards().copy())
        outs.append((np.array(rs), [eng.dump(a) for a in range(A)], eng.dones().copy()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][2], outs[1][2])
    for b0, b1 in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(b0, b1)


@pytest.mark.parametrize("qg,fused,tile", [(1, 0, 0), (2, 0, 0), (4, 0, 0), (8, 0, 0), (16, 0, 0), (1, 1, 6), (2, 1, 0), (4, 1, 6), (8, 1, 0), (2, 0, 6)])
def test_front_kernel_lanes_per_arena(hip_engine_cls, oracle_lib, monkeypatch, qg, fused, tile):
    """The lean front part runs with 1, 2, 4, 8 or 16 lanes per arena (chosen from the arena count; AGARCL_QUIET_QG / AGARCL_FUSED_QG
    pin it, AGARCL_FUSED selects the two-kernel or the single-launch step, AGARCL_TILE_LG the layout of the word arrays): 64 / 32 /
    16 / 8 / 4 arenas share a wavefront and its pellet passes.  Quiet C2 arenas plus a few big ones, lock-step against the oracle,
    rewards every step and whole state at the end."""
    from oracle import blob
    monkeypatch.setenv("AGARCL_QUIET_QG", str(qg)); monkeypatch.setenv("AGARCL_FUSED_QG", str(qg))
    monkeypatch.setenv("AGARCL_FUSED", str(fused)); monkeypatch.setenv("AGARCL_TILE_LG", str(tile))
    A, steps = 150, 250   # 150 arenas: the last wavefront is ragged for every group size
    eng = hip_engine_cls(A, **C2)
    oras = [oracle_lib.OraEnv(**C2) for _ in range(A)]
    seeds = (31000 + 7 * np.arange(A)).astype(np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for a, o in enumerate(oras):
        o.seed(int(seeds[a])); o.reset(True)
        if a % 37 == 5:   # a big cell: every tick eats, so the front part keeps handing over
            d = blob.parse(o.dump()); d["players"][0]["cell_mass"][0] = 900
            b = blob.build(d); o.load(b); eng.load(b, a)
    rng = np.random.RandomState(qg + 100 * fused + tile)
    for t in range(steps):
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = np.zeros((A, 1), dtype=np.int32)
        eng.set_actions(dxdy, act); eng.step()
        r = eng.rewards()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); ro = oras[a].step()
            assert r[a, 0] == ro[0], (t, a)
    for a in range(A):
        assert blob.diff(oras[a].dump(), eng.dump(a)) is None, a
    assert not eng.flags().any()
    eng.close()


@pytest.mark.parametrize("cfg", [C2, dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6),
                                 dict(num_agents=2, arena_size=250, num_pellets=500, num_viruses=10, num_bots=3, mode=0)])
def test_tile_transposed_layout(hip_engine_cls, oracle_lib, monkeypatch, cfg):
    """Per-arena word arrays in tiles of 64 arenas (the layout of single-player batches from 32768 arenas on; AGARCL_TILE_LG=6
    pins it): lock-step against the oracle through the front kernel (4 lanes per arena), the general engine and a multi-player
    config; 130 arenas = two full tiles and a ragged one."""
    monkeypatch.setenv("AGARCL_TILE_LG", "6"); monkeypatch.setenv("AGARCL_QUIET_QG", "4"); monkeypatch.setenv("AGARCL_FUSED", "0")
    A = 130
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 160, seeds=7000 + 3 * np.arange(A), policy_seed=3, sticky=4, every=20)
    eng.close()
    assert ok, msg


from snapshot_cases import replay_snapshot_case, snapshot_cases  # noqa: E402


@pytest.mark.parametrize("base", snapshot_cases(), ids=lambda p: os.path.basename(p))
def test_hip_snapshot_golden(hip_engine_cls, base):
    """a snapshot written by the real reference is loaded into one arena of the HIP engine, which then
    follows the reference's recorded continuation bit for bit and re-serialises to the reference's own JSON."""
    ok, msg = replay_snapshot_case(hip_engine_cls, base)
    assert ok, msg


def test_baseline_config_c1_population(hip_engine_cls, oracle_lib):
    """BASELINE.json configs[0] / SURVEY 8(d) C1 -- the reference's own CPU-runnable case: 250x250 arena, 500 pellets,
    10 viruses, one agent + the four bot kinds, random targets and actions, dead players respawned (mode 0), 10 000
    ticks -- in lock-step with the oracle on the GPU, driven through the env API (whose tick length is the reference's
    fixed 1/30 s; bench/main.cpp's engine-level 1/60 s variant is test_engine_level_60hz_bench_path below)."""
    cfg = dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0)
    A = 4
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 2500, seeds=np.array([42, 43, 44, 45]), sticky=1, every=50)
    eng.close()
    assert ok, msg


def test_engine_level_60hz_bench_path(hip_engine_cls, oracle_lib):
    """BASELINE configs[0] exactly as bench/main.cpp:14-38 drives it: Engine::tick at dt = 1/60 s (600-tick recombine
    deadlines) through agarcl_set_targets / agarcl_tick, agent + the four bot kinds, dead players respawned every tick;
    then a lone mass-1000 agent whose split cells wait out the full 600-tick deadline."""
    cfg = dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
    A = 3
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_engine_level_lockstep(eng, oras, 2000, seeds=[42, 43, 44], every=20)
    eng.close()
    assert ok, msg
    cfg = dict(num_agents=1, arena_size=300, num_pellets=300, num_viruses=5, mode=6, dt=1.0 / 60)
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_engine_level_lockstep(eng, oras, 1500, seeds=[5, 6, 7], every=20, respawn=False)
    eng.close()
    assert ok, msg


def test_masked_reset_device_mask(hip_engine_cls, oracle_lib):
    """agarcl_reset_device: the mask stays in HBM (here: a torch tensor), the reset is a stream-ordered launch"""
    import torch
    from oracle import blob
    A = 6
    eng = hip_engine_cls(A, **C3)
    oras = [oracle_lib.OraEnv(**C3) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 20, seeds=np.arange(60, 60 + A), sticky=4, every=20)
    assert ok, msg
    mask = torch.tensor([0, 1, 1, 0, 0, 1], dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.reset_device(mask.data_ptr(), reset_ids=False)
    for a in range(A):
        if int(mask[a]):
            oras[a].reset(False)
        assert blob.diff(oras[a].dump(), eng.dump(a)) is None, "arena %d after the device-masked reset" % a
    eng.close()


def test_flag_watch_reports_overflow(hip_engine_cls):
    """A capacity overflow (here: more foods than cap_foods) raises the arena's sticky flag AND reaches the host through the
    asynchronous watch word without a synchronising query (agarcl_poll_flags)."""
    A = 8
    eng = hip_engine_cls(A, arena_size=200, num_pellets=100, num_viruses=0, mode=6, cap_foods=4)
    eng.seed(None, 3); eng.reset(reset_ids=True)
    assert eng.poll_flags() == 0
    dxdy = np.tile(np.array([[[0.7, 0.1]]], np.float32), (A, 1, 1)); act = np.ones((A, 1), np.int32)     # feed, feed, feed
    for t in range(200):
        eng.set_actions(dxdy, act); eng.step()
    assert (eng.flags() & 2).any()             # AGARCL_F_FOODS_OVERFLOW
    eng.sync()
    for t in range(70):                        # the watch samples every 64 steps
        eng.step()
    eng.sync(); eng.step()
    assert eng.poll_flags() & 2
    eng.close()


def test_masked_reset_and_reseed(hip_engine_cls, oracle_lib):
    """reset(mask) touches only the selected arenas; ids keep growing like the reference's global counter."""
    A = 8
    eng = hip_engine_cls(A, **C3M6)
    oras = [oracle_lib.OraEnv(**C3M6) for _ in range(A)]
    seeds = np.arange(40, 40 + A).astype(np.uint32)
    ok, msg = run_batched_lockstep(eng, oras, 40, seeds=seeds, sticky=8, every=40)
    assert ok, msg
    mask = np.array([1, 0, 0, 1, 0, 1, 0, 0], dtype=np.uint8)
    eng.reset(mask, reset_ids=False)
    for a in range(A):
        if mask[a]:
            oras[a].reset(False)
    from oracle import blob
    for a in range(A):
        assert blob.diff(oras[a].dump(), eng.dump(a)) is None, "arena %d after masked reset" % a
    for t in range(40):
        dxdy = np.zeros((A, 1, 2), np.float32); act = np.zeros((A, 1), np.int32)
        for a in range(A):
            dd, aa = policy(9 + a, t, 1, True, 8); dxdy[a, 0] = dd[0]; act[a, 0] = aa[0]
        eng.set_actions(dxdy, act); eng.step()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); oras[a].step()
    for a in range(A):
        assert blob.diff(oras[a].dump(), eng.dump(a)) is None, "arena %d after continuing" % a
    eng.close()


def test_events_match_oracle(hip_engine_cls, oracle_lib):
    """eat events (pellets_to_remove / viruses_to_remove order, Engine.hpp:992,1243) bit-exact per tick."""
    cfg = dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6)
    A = 8
    eng = hip_engine_cls(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    seeds = np.arange(70, 70 + A).astype(np.uint32)
    eng.seed(seeds); eng.reset(reset_ids=True)
    for o, s in zip(oras, seeds):
        o.seed(int(s)); o.reset(True)
    total = 0
    for t in range(200):
        eng.tick(1)
        n, pe, ve = eng.events()
        for a in range(A):
            oras[a].tick()
            ope, ove = oras[a].last_events()
            assert np.array_equal(ope, pe[a, :n[a, 0]]), (t, a)
            assert np.array_equal(ove, ve[a, :n[a, 1]]), (t, a)
            total += len(ope) + len(ove)
    assert total > 50
    eng.close()


def test_full_size_properties_4096(hip_engine_cls, oracle_lib):
    """BASELINE config 2 at full size (4096 arenas): (i) a sample of arenas equals the oracle run alone on
    the same seed (batch independence), (ii) two identical runs are bitwise identical (determinism),
    (iii) conservation: without decay-free ... mass gained == pellets eaten, pellet count <= target,
    entity ids unique and below the id counter."""
    from oracle import blob
    A, steps = 4096, 50
    rng = np.random.RandomState(3)
    dx = rng.uniform(-1, 1, size=(steps, A, 1, 2)).astype(np.float32)
    act = np.zeros((A, 1), np.int32)
    runs = []
    for rep in range(2):
        eng = hip_engine_cls(A, **C2)
        eng.seed(None, 10000); eng.reset(reset_ids=True)
        tot_reward = np.zeros((A, 1))
        for t in range(steps):
            eng.set_actions(dx[t], act); eng.step()
            tot_reward += eng.rewards()
        assert not eng.flags().any()
        sample = [0, 1, 63, 64, 1000, 2047, 4095]
        runs.append(([eng.dump(a) for a in sample], eng.masses().copy(), eng.counts().copy(), tot_reward))
        if rep == 0:
            for a, b in zip(sample, runs[0][0]):
                o = oracle_lib.OraEnv(**C2); o.seed(10000 + a); o.reset(True)
                for t in range(steps):
                    o.take_actions(dx[t, a], act[a]); o.step()
                assert blob.diff(o.dump(), b) is None, "arena %d differs from the oracle run alone" % a
                d = blob.parse(b)
                ids = np.concatenate([d["pellet_id"], d["virus_id"], d["food_id"], d["players"][0]["cell_id"]])
                assert len(np.unique(ids)) == len(ids) and ids.max() <= d["id_counter"]
        eng.close()
    for b0, b1 in zip(runs[0][0], runs[1][0]):
        assert np.array_equal(b0, b1)
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    masses, counts, tot_reward = runs[0][1], runs[0][2], runs[0][3]
    assert (counts[:, 0] <= 1000).all() and (counts[:, 3] == 1).all()
    # reward_type 1: the rewards of an episode sum to (final mass - initial mass)
    assert np.array_equal(tot_reward[:, 0], masses[:, 0] - 25.0)


def test_mode6_full_size_vs_oracle_sample(hip_engine_cls, oracle_lib):
    """BASELINE config 3 (full ruleset) at 4096 arenas: sampled arenas vs the oracle after 60 steps."""
    from oracle import blob
    A, steps = 4096, 60
    eng = hip_engine_cls(A, **C3M6)
    eng.seed(None, 777); eng.reset(reset_ids=True)
    acts = []
    for t in range(steps):
        rng = np.random.RandomState(1000 + t // 8)
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32)
        a = rng.randint(0, 3, size=(A, 1)).astype(np.int32)
        acts.append((dxdy, a))
        eng.set_actions(dxdy, a); eng.step()
    assert not eng.flags().any()
    for arena in (0, 5, 777, 4095):
        o = oracle_lib.OraEnv(**C3M6); o.seed(777 + arena); o.reset(True)
        for dxdy, a in acts:
            o.take_actions(dxdy[arena], a[arena]); o.step()
        assert blob.diff(o.dump(), eng.dump(arena)) is None, "arena %d" % arena
    eng.close()


def test_config4_single_gpu_half_32768_arenas_mode6(hip_engine_cls, oracle_lib):
    """BASELINE configs[3]'s per-node size on ONE GPU: 32 768 arenas of the full ruleset (mode 6); sampled arenas equal the
    oracle run alone on the same seed and action stream, no capacity flag anywhere."""
    from oracle import blob
    A, steps = 32768, 40
    eng = hip_engine_cls(A, **C3M6)
    eng.seed(None, 31000); eng.reset(reset_ids=True)
    acts = []
    for t in range(steps):
        rng = np.random.RandomState(2000 + t // 8)
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32)
        a = rng.randint(0, 3, size=(A, 1)).astype(np.int32)
        acts.append((dxdy, a))
        eng.set_actions(dxdy, a); eng.step()
    assert not eng.flags().any()
    counts = eng.counts()
    assert (counts[:, 3] >= 1).all() and counts[:, 3].mean() > 3       # the agents really split (full ruleset at work)
    for arena in (0, 1, 4095, 4096, 20011, 32767):
        o = oracle_lib.OraEnv(**C3M6); o.seed(31000 + arena); o.reset(True)
        for dxdy, a in acts:
            o.take_actions(dxdy[arena], a[arena]); o.step()
        assert blob.diff(o.dump(), eng.dump(arena)) is None, "arena %d" % arena
    eng.close()


@pytest.mark.parametrize("A", [65536, 150000])
def test_big_quiet_batches_default_policy(hip_engine_cls, oracle_lib, A):
    """North-star batch sizes on ONE GPU with the engine's own choices (65 536 arenas: single launch, 2 lanes per arena, tiled
    word arrays; 150 000: two kernels, 1 lane per arena, ragged last tile): C2, 60 steps; sampled arenas -- first / last of
    tiles and wavefronts, the ragged tail -- equal the oracle run alone on the same seed and action stream; no flag anywhere;
    every arena-step accounted for by the work counters."""
    from oracle import blob
    steps = 60
    eng = hip_engine_cls(A, **C2)
    eng.seed(None, 52000); eng.reset(reset_ids=True)
    eng.work(reset=True)
    sample = sorted({0, 1, 31, 32, 63, 64, 65, 4095, 4096, A // 2 + 7, A - 65, A - 64, A - 2, A - 1})
    acts = []
    for t in range(steps):
        rng = np.random.RandomState(7000 + t // 4)
        dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32)
        a = np.zeros((A, 1), dtype=np.int32)
        acts.append((dxdy[sample].copy(), a[sample].copy()))
        eng.set_actions(dxdy, a); eng.step()
    assert not eng.flags().any()
    w = eng.work()
    assert int(w[0]) + int(w[1]) == A * steps and int(w[0]) > 0.99 * A * steps
    for k, arena in enumerate(sample):
        o = oracle_lib.OraEnv(**C2); o.seed(52000 + arena); o.reset(True)
        for dxdy, a in acts:
            o.take_actions(dxdy[k], a[k]); o.step()
        assert blob.diff(o.dump(), eng.dump(arena)) is None, "arena %d" % arena
    eng.close()


def test_step_actions_is_take_actions_plus_step(hip_engine_cls):
    """agarcl_step_actions (one host call per env step, policy output already in HBM) == agarcl_set_actions(on_device) + agarcl_step."""
    import torch
    A, steps = 96, 40
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(5)
    dxdy = (torch.rand((steps, A, 1, 2), generator=g, device=dev) * 2 - 1).contiguous()
    act = torch.randint(0, 3, (steps, A, 1), generator=g, device=dev, dtype=torch.int32)
    outs = []
    for fused_call in (False, True):
        eng = hip_engine_cls(A, **C3)
        eng.seed(None, 777); eng.reset(reset_ids=True)
        rs = []
        for t in range(steps):
            if fused_call:
                eng.step_actions(dxdy[t].data_ptr(), act[t].data_ptr(), 0)
            else:
                eng.set_actions_device(dxdy[t].data_ptr(), act[t].data_ptr()); eng.step()
            rs.append(eng.rewards().copy())
        outs.append((np.array(rs), [eng.dump(a) for a in range(A)]))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for b0, b1 in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(b0, b1)


def test_stream_timer(hip_engine_cls):
    """agarcl_timer_mark / agarcl_timer_elapsed_ms: events on the env's own stream bracket the steps launched between them."""
    eng = hip_engine_cls(256, **C2)
    eng.seed(None, 1); eng.reset(reset_ids=True)
    eng.timer_mark(0)
    for _ in range(50):
        eng.step()
    eng.timer_mark(1)
    ms = eng.timer_elapsed_ms()
    assert 0.05 < ms < 50.0, ms          # 50 steps of a few microseconds each
    eng.close()


def test_mode3_done_threshold(hip_engine_cls, oracle_lib):
    """Mode 3 ends an episode when the agent's mass reaches 23 000 (BaseEnvironment.hpp:108-111; pinned against the reference in
    test_oracle_vs_reference.py): cells loaded just below the threshold cross it by eating -- through the lean front part (arena 0:
    sparse pellets) and through the general engine (crowded arenas) -- dones, rewards and state against the oracle."""
    from oracle import blob
    cfg = dict(num_agents=1, arena_size=300, num_pellets=600, num_viruses=0, mode=3, reward_type=1)
    A = 4
    eng = hip_engine_cls(A, **cfg)
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


def test_negative_decay_factor_wraps(hip_engine_cls, oracle_lib):
    """Entities.hpp:199-202 in the reference's x86-64 build: with >= 66 virus meals inside the anti-team window the decay factor
    1 - 0.002 * 1.1^k is negative and the double -> uint32 conversion wraps (mass 2^32 - x, not 0).  A player loaded with 70 recent
    virus meals reaches its next decay check: the same wrapped mass as the reference (state equal right after that step), and -- the
    mass now exceeds the radius / speed tables -- the capacity flag instead of a silent divergence."""
    from oracle import blob
    cfg = dict(num_agents=1, arena_size=300, num_pellets=100, num_viruses=0, mode=0, reward_type=1)
    A = 3
    eng = hip_engine_cls(A, **cfg)
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
def test_wrapped_mass_terminates(hip_engine_cls):
    """A mass that has wrapped around 2^32 (the reference's unsigned arithmetic allows it: negative decay factor) must not hang the
    growth closure of the pellet scan (m + K wraps back to a tiny mass and the candidate count would oscillate): the step ends
    and the arena carries AGARCL_F_MASS_LUT_OVERFLOW."""
    from oracle import blob
    cfg = dict(num_agents=1, arena_size=80, num_pellets=64, num_viruses=0, mode=0)
    A = 2
    eng = hip_engine_cls(A, **cfg)
    eng.seed(np.array([5, 6], dtype=np.uint32)); eng.reset(reset_ids=True)
    for a in range(A):
        d = blob.parse(eng.dump(a)); d["players"][0]["cell_mass"][0] = (1 << 32) - 3 - a
        eng.load(blob.build(d), a)
    for t in range(8):
        eng.set_actions(np.zeros((A, 1, 2), np.float32), np.zeros((A, 1), np.int32)); eng.step()
    assert (eng.flags() & 32).all()
    eng.close()


def test_mass_beyond_tables_is_flagged(hip_engine_cls):
    """The reference computes radius and speed from the mass without a limit; the tables end at 2^19.  A cell beyond that makes every
    look-up clamp, so the arena must carry AGARCL_F_MASS_LUT_OVERFLOW after the very next step -- through the lean front part (arena 0:
    one quiet cell) and through the general engine (arena 1: viruses around, several cells after a split)."""
    from oracle import blob
    cfg = dict(num_agents=1, arena_size=400, num_pellets=200, num_viruses=5, mode=0)
    A = 3
    eng = hip_engine_cls(A, **cfg)
    eng.seed(np.array([5, 6, 7], dtype=np.uint32)); eng.reset(reset_ids=True)
    for a in (0, 1):
        d = blob.parse(eng.dump(a)); d["players"][0]["cell_mass"][0] = 600000 + a
        if a == 0: d["virus_x"] = d["virus_x"][:0]; d["virus_y"] = d["virus_y"][:0]; d["virus_vx"] = d["virus_vx"][:0]; d["virus_vy"] = d["virus_vy"][:0]; d["virus_mass"] = d["virus_mass"][:0]; d["virus_hits"] = d["virus_hits"][:0]; d["virus_id"] = d["virus_id"][:0]
        eng.load(blob.build(d), a)
    eng.set_actions(np.zeros((A, 1, 2), np.float32), np.array([[0], [2], [0]], np.int32)); eng.step()
    fl = eng.flags()
    assert fl[0] & 32 and fl[1] & 32 and fl[2] == 0, fl
    eng.close()


def test_error_paths(hip_engine_cls):
    from agarcl_amd._capi import AgarclError
    with pytest.raises(AgarclError):
        hip_engine_cls(1, mode=11)                      # Engine.hpp:413-414 "Invalid mode number"
    with pytest.raises(AgarclError):
        hip_engine_cls(1, num_agents=12, num_bots=8, example_bots=13, mode=0)   # more than 32 players per arena: loud, not a silent fallback
    with pytest.raises(AgarclError):
        hip_engine_cls(1, num_pellets=5000)             # beyond the pellet register file
    e = hip_engine_cls(2, **C2)
    with pytest.raises(Exception):
        e.set_actions(np.zeros((3, 1, 2), np.float32), np.zeros((3, 1), np.int32))
    e.close()


def test_vec_environment_device_reset_and_flag_watch(oracle_lib):
    """VecEnvironment (torch tensors in and out): finished arenas are restarted with a CUDA mask (stream-ordered, no host
    round trip) and a capacity overflow surfaces as an error from step() through the asynchronous flag watch."""
    import torch
    from agarcl_amd.vec_env import VecEnvironment
    from agarcl_amd._capi import AgarclError
    from oracle import blob
    A = 32
    env = VecEnvironment(A, num_viruses=5, mode_number=0, arena_size=300, num_pellets=300)
    env.seed(base_seed=77); env.reset(reset_ids=True)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    for t in range(30):
        dxdy = torch.rand((A, 1, 2), generator=g, device=env.device) * 2 - 1
        env.take_actions(dxdy, torch.zeros((A, 1), dtype=torch.int32, device=env.device)); env.step()
    mask = (torch.arange(A, device=env.device) % 3 == 0).to(torch.uint8)
    before = [env.engine.dump(a) for a in range(A)]
    env.reset(mask)                                   # device mask: agarcl_reset_device
    env.sync()
    for a in range(A):
        now = env.engine.dump(a)
        if a % 3 == 0:
            assert blob.parse(now)["ticks"] == 0 and not np.array_equal(now, before[a])
        else:
            assert np.array_equal(now, before[a])
    env.close()
    env = VecEnvironment(8, arena_size=200, num_pellets=100, mode_number=6, cap_foods=4)     # 4 food slots: feeding overflows
    env.seed(base_seed=3); env.reset(reset_ids=True)
    dxdy = torch.full((8, 1, 2), 0.5, device=env.device); act = torch.ones((8, 1), dtype=torch.int32, device=env.device)
    with pytest.raises(AgarclError):
        for t in range(400):
            env.take_actions(dxdy, act); env.step()
            if t % 50 == 49:
                env.sync()
    env.close()


def test_flag_watch_restarts_on_masked_reset():
    """The normal RL auto-reset path: an arena raises a capacity flag, step() reports it (strict_flags), the caller resets exactly the
    flagged arenas with a device mask -- and the watch must be clean again: 130 further steps without an error, poll_flags() == 0.
    A masked reset that leaves a flagged arena alone keeps reporting it."""
    import torch
    from agarcl_amd.vec_env import VecEnvironment
    from agarcl_amd._capi import AgarclError
    A = 8
    env = VecEnvironment(A, arena_size=200, num_pellets=100, mode_number=6, cap_foods=4)     # 4 food slots: feeding overflows
    env.seed(base_seed=3); env.reset(reset_ids=True)
    feed = torch.full((A, 1, 2), 0.5, device=env.device), torch.ones((A, 1), dtype=torch.int32, device=env.device)
    idle = torch.zeros((A, 1, 2), device=env.device), torch.zeros((A, 1), dtype=torch.int32, device=env.device)
    with pytest.raises(AgarclError):
        for t in range(400):
            env.take_actions(*feed); env.step()
            if t % 50 == 49:
                env.sync()
    env.sync()
    flagged = env.engine.flags() != 0
    assert flagged.any()
    # (1) a masked reset that misses one flagged arena: the watch still reports
    first = int(np.flatnonzero(flagged)[0])
    part = flagged.copy(); part[first] = False
    env.reset(torch.as_tensor(part.astype(np.uint8), device=env.device))
    env.sync()
    assert (env.engine.flags() != 0).tolist() == [a == first for a in range(A)]
    with pytest.raises(AgarclError):
        for t in range(200):
            env.take_actions(*idle); env.step()
            if t % 50 == 49:
                env.sync()
    # (2) the reset the error message asks for: every flagged arena
    env.sync()
    env.reset(torch.as_tensor((env.engine.flags() != 0).astype(np.uint8), device=env.device))
    for t in range(130):
        env.take_actions(*idle); env.step()     # raises if the watch is sticky
        if t % 40 == 39:
            env.sync()
    env.sync()
    assert env.engine.poll_flags() == 0 and not env.engine.flags().any()
    # (3) the same with a host mask
    with pytest.raises(AgarclError):
        for t in range(400):
            env.take_actions(*feed); env.step()
            if t % 50 == 49:
                env.sync()
    env.sync()
    env.reset((env.engine.flags() != 0).astype(np.uint8))
    for t in range(130):
        env.take_actions(*idle); env.step()
        if t % 40 == 39:
            env.sync()
    env.sync()
    assert env.engine.poll_flags() == 0
    env.close()


def test_grid_obs_plain_call_voids_the_undo_list(hip_engine_cls):
    """agarcl_grid_obs(on_device=2, buf), (on_device=1, buf), (on_device=2, buf): the middle call's scattered words are not in the undo
    list of the first, so the third must clear everything -- its contents equal a fresh full-clear observation."""
    import torch
    A, G = 6, 32
    eng = hip_engine_cls(A, **C3M6)
    eng.seed(None, 9); eng.reset(reset_ids=True)
    rng = np.random.RandomState(4)
    dev = torch.device("cuda", 0)
    C = 8   # 1 + cells + 2 others + 2 viruses + 2 pellets
    buf = torch.full((A, 1, C, G, G), 7, dtype=torch.int32, device=dev); ref = torch.empty_like(buf)
    def step(n):
        for _ in range(n):
            eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32)); eng.step()
    def obs(t, mode):
        assert eng.grid_obs(G, out_ptr=t.data_ptr(), persistent=(mode == 2)) == C
        eng.sync()
    step(5); obs(buf, 2)
    step(5); obs(buf, 1)
    step(5); obs(buf, 2); obs(ref, 1)
    assert torch.equal(buf, ref)
    step(5); obs(buf, 2); obs(ref, 1)           # and the incremental path proper still agrees
    assert torch.equal(buf, ref)
    eng.close()


def test_work_counters_account_for_every_arena_step(hip_engine_cls):
    """agarcl_debug_work (what bench.py's requested-bytes figure is built on): every arena-step is counted exactly once, as
    finished by the lean front part or as having gone through the general engine; pellet-array transfers are counted."""
    A, steps = 512, 64
    for cfg, mostly_front in ((C2, True), (C3M6, False)):
        eng = hip_engine_cls(A, **cfg)
        eng.seed(None, 1); eng.reset(reset_ids=True)
        rng = np.random.RandomState(0)
        eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), np.zeros((A, 1), np.int32)); eng.step()   # first step
        eng.work(reset=True)
        for t in range(steps):
            eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32) if not mostly_front else np.zeros((A, 1), np.int32))
            eng.step()
        w = eng.work()
        assert w[0] + w[1] == A * steps, (cfg, w)
        assert w[2] > 0
        if mostly_front:
            assert w[0] > 0.99 * A * steps
        else:
            assert w[1] > 0.9 * A * steps and w[2] >= A * steps    # mass-1000 agents: the general engine every step, pellets read every launch
        eng.close()


@pytest.mark.parametrize("trial", [(91, 7), (62, 28), None], ids=["soak91_7", "soak62_28", "c3m6"])
@pytest.mark.parametrize("qg", [16, 4])
def test_fused_general_tail_regression(hip_engine_cls, oracle_lib, monkeypatch, trial, qg):
    """k_fused pinned, tiled layout, 130 arenas in a mass-1000 mode: every arena-step runs general_arena_step behind the front part,
    several arenas per wavefront one after the other, incl. the workgroup's first wavefront (LDS block at offset 0).
    (91, 7) and (62, 28) are the scripts/gpu_soak.py trials that killed the queue in round 2 (memory-aperture violation: flat_* LDS access
    whose folded offset left the address register below the aperture base), replayed with their own seeds and policy;
    the third case is the shipped default shape (16 pellet slots, 2 x 2 pellet grid)."""
    from lockstep import soak_trial
    if trial is None:
        cfg, A, seeds, ps, st = dict(C3M6), 130, np.arange(4242, 4242 + 130), 5, 4
    else:
        cfg, _, A, seeds, ps, st = soak_trial(*trial)
        assert (cfg["arena_size"], cfg["num_pellets"], cfg["mode"], A) == (1100, 1300, 6, 130)
    monkeypatch.setenv("AGARCL_FUSED", "1"); monkeypatch.setenv("AGARCL_TILE_LG", "6"); monkeypatch.setenv("AGARCL_FUSED_QG", str(qg))
    eng = hip_engine_cls(A, **cfg)
    assert eng.L.agarcl_debug_fused(eng.h) == 1, "the single-launch step must serve this env (no fence on pellet slots / pellet grid)"
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 120, seeds=np.asarray(seeds, dtype=np.uint32), policy_seed=ps, sticky=st, every=30)
    eng.close()
    assert ok, msg
