"""The tracked pellet of the quiet path (agar_core.inl quiet_ticks, PL_CAND_*): a pellet pass names the nearest pellet, the pellet-free disc
reaches to the SECOND nearest one, and eating the tracked pellet needs no further pass.  Hand-made arenas in which exactly that has to happen
-- a cell walking straight onto a lone pellet, two equally near pellets (the tie goes to the lower index), a tracked pellet that is the
array's last one, an arena with a single pellet -- tick for tick against the C oracle (Engine.hpp:976-1009), on the kernel source (host emulation) and on
the HIP engine; the pass counter (PL_PASSES) proves that the eat happened WITHOUT a pass."""
import numpy as np
import pytest

from oracle import blob

CFG = dict(num_agents=1, ticks_per_step=1, arena_size=1000, num_pellets=1000, num_viruses=0, mode=3, pellet_regen=False)   # mode 3: no decay
PL_FOOD_EATEN, PL_PASSES, PL_CAND_IDX, AR_SAFE = 9, 19, 22, 47   # agar_types.h


def arena_with(o, cell_xy, pellets, mass=25):
    d = blob.parse(o.dump()); pl = d["players"][0]
    pl["cell_f"] = np.array([[cell_xy[0], cell_xy[1], 0, 0, 0, 0]], dtype=np.float32)
    pl["cell_mass"] = np.array([mass], dtype=np.int64); pl["cell_id"] = np.array([7], dtype=np.int64); pl["cell_recomb"] = np.array([0], dtype=np.int64)
    pl["n_cells"] = 1; pl["highest_mass"] = mass; pl["min_mass_cell"] = mass
    p = np.asarray(pellets, dtype=np.float32).reshape(-1, 2)
    d["pellet_x"], d["pellet_y"], d["pellet_id"] = p[:, 0].copy(), p[:, 1].copy(), (100 + np.arange(len(p))).astype(np.int64)
    d["ticks"] = 1          # the next regeneration tick (ticks % 120 == 0, Engine.hpp:236-239) lies beyond the test
    return blob.build(d)


# (name, cell, pellets, direction): every case keeps the cell inside the disc of its first pass until it has eaten
CASES = [
    ("walk onto a lone near pellet, the others far away", (500, 500), [(900, 900), (530, 500), (100, 900), (900, 100)], (1, 0)),
    ("two equally near pellets left and right: the lower index is tracked, the walk goes to the other one", (500, 500), [(470, 500), (530, 500), (900, 900)], (1, 0)),
    ("two equally near pellets, the walk goes to the tracked one", (500, 500), [(470, 500), (530, 500), (900, 900)], (-1, 0)),
    ("the tracked pellet is the last of the array (no swap on removal)", (300, 300), [(900, 900), (100, 900), (300, 330)], (0, 1)),
    ("one pellet only", (300, 300), [(300, 335)], (0, 1)),
    ("a diagonal walk that passes the tracked pellet within the grown radius only", (500, 500), [(520, 517), (900, 100), (100, 100)], (1, 0.72)),
]


def run_cases(make_engine, oracle_lib, ticks=70):
    A = len(CASES)
    eng = make_engine(A, **CFG)
    oras = [oracle_lib.OraEnv(**CFG) for _ in range(A)]
    eng.seed(np.arange(40, 40 + A, dtype=np.uint32)); eng.reset(reset_ids=True)
    for a, (o, (_, cell, pellets, _)) in enumerate(zip(oras, CASES)):
        o.seed(40 + a); o.reset(True)
        b = arena_with(o, cell, pellets)
        o.load(b); eng.load(b, a)
    dxdy = np.array([[c[3]] for c in CASES], dtype=np.float32).reshape(A, 1, 2); act = np.zeros((A, 1), np.int32)
    at_track = [None] * A     # (passes, eaten) when a disc with a tracked pellet was first in force
    for t in range(ticks):
        eng.set_actions(dxdy, act); eng.step()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); oras[a].step()
            d = blob.diff(oras[a].dump(), eng.dump(a))
            assert d is None, "tick %d, case '%s': %s" % (t, CASES[a][0], d)
            ar, pl = eng.arena_words(a); w = pl[0]
            if at_track[a] is None and w[PL_CAND_IDX] >= 0 and ar[AR_SAFE] != 0:   # (the tracked pellet counts only inside a disc: AR_SAFE > 0)
                at_track[a] = (int(w[PL_PASSES]), int(w[PL_FOOD_EATEN]))
    end = [(int(w[PL_PASSES]), int(w[PL_FOOD_EATEN])) for w in (eng.arena_words(a)[1][0] for a in range(A))]
    fl = eng.flags(); eng.close()
    assert not fl.any(), fl
    return at_track, end


TIES = (1, 2)   # equally near pellets: the disc ends at the nearest one as it did before, the walk needs a second pass


def check(at_track, end):
    for a, (name, _, _, _) in enumerate(CASES):
        assert at_track[a] is not None, "case '%s': no pellet was ever tracked" % name
        assert at_track[a][1] == 0 and end[a][1] == 1, "case '%s': eaten %r -> %r" % (name, at_track[a], end[a])
        if a not in TIES:   # the pellet was eaten on the pass that built the disc: no pass since
            assert end[a][0] == at_track[a][0], "case '%s': passes %d -> %d" % (name, at_track[a][0], end[a][0])


def test_tracked_pellet_cases_emulated(emu_lib, oracle_lib):
    from agarcl_amd import _capi
    check(*run_cases(lambda A, **cfg: _capi.BatchedEngine(A, lib=emu_lib, **cfg), oracle_lib))


@pytest.mark.gpu
def test_tracked_pellet_cases_hip(hip_engine_cls, oracle_lib, monkeypatch):
    # the single-launch step pinned: with the adaptive choice the first steps after a state load may run on k_step alone, whose launches
    # count their own pellet-array load in PL_PASSES
    monkeypatch.setenv("AGARCL_FUSED", "1")
    check(*run_cases(hip_engine_cls, oracle_lib))


def run_pass_budget(make_engine, oracle_lib, steps=500):
    """C2 under its own policy: bit-exact, and the pellet-array transfers (the reset's included) stay below one per 30 arena-steps (one per
    18 before the pellet was tracked)."""
    from lockstep import run_quiet_rollout
    cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
    A = 8
    eng = make_engine(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_quiet_rollout(eng, oras, steps, 52000 + np.arange(A), rng_seed=9, check_every=50)
    assert ok, msg
    passes = sum(int(eng.arena_words(a)[1][0][PL_PASSES]) for a in range(A)); eaten = sum(int(eng.arena_words(a)[1][0][PL_FOOD_EATEN]) for a in range(A))
    eng.close()
    assert eaten > 0
    assert passes * 30 < A * steps, (passes, A * steps)


def test_pass_budget_emulated(emu_lib, oracle_lib):
    from agarcl_amd import _capi
    run_pass_budget(lambda A, **cfg: _capi.BatchedEngine(A, lib=emu_lib, **cfg), oracle_lib)


@pytest.mark.gpu
def test_pass_budget_hip(hip_engine_cls, oracle_lib):
    run_pass_budget(hip_engine_cls, oracle_lib)
