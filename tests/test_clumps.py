"""Self-collision relaxation on loaded states with MANY cells in one dense clump (Engine.hpp:763-938).  Play never makes
more than 14-16 cells per player, but a loaded state may hold up to the cell capacity (32), and the level walk has code that only such
states reach: the second and later rounds of the touch-bit pass (more than 64 / 128 pairs), the upper half of the level mask (18+ cells),
phases with D == LL (2 and 3 cells).  The kernel source on the host emulation and the HIP engine against the C oracle (itself pinned
against the reference), tick by tick."""
import numpy as np
import pytest

from lockstep import policy
from oracle import blob


def clump_state(o, n, seed, spread=6.0, masses=(30, 400)):
    rng = np.random.RandomState(seed)
    d = blob.parse(o.dump()); pl = d["players"][0]
    xy = 150 + rng.uniform(-spread, spread, size=(n, 2))
    pl["cell_f"] = np.concatenate([xy, np.zeros((n, 4))], axis=1).astype(np.float32)
    pl["cell_mass"] = rng.randint(masses[0], masses[1], size=n).astype(np.int64)
    pl["cell_id"] = (9000 + 10 * np.arange(n)).astype(np.int64)
    pl["cell_recomb"] = np.full(n, 300, dtype=np.int64)          # nobody recombines during the test
    pl["n_cells"] = n
    pl["highest_mass"] = max(int(pl["highest_mass"]), int(pl["cell_mass"].sum()))
    return blob.build(d)


def run_clumps(make_engine, oracle_lib, ticks=24):
    cfg = dict(num_agents=1, ticks_per_step=1, arena_size=300, num_pellets=60, num_viruses=0, mode=3)   # mode 3: no decay
    sizes = [2, 3, 4, 7, 11, 12, 14, 16, 17, 18, 23, 32]
    A = len(sizes)
    eng = make_engine(A, **cfg)
    oras = [oracle_lib.OraEnv(**cfg) for _ in range(A)]
    eng.seed(np.arange(70, 70 + A, dtype=np.uint32)); eng.reset(reset_ids=True)
    for a, (o, n) in enumerate(zip(oras, sizes)):
        o.seed(70 + a); o.reset(True)
        b = clump_state(o, n, seed=5 * a + 1, spread=3.0 + 0.4 * n)
        o.load(b); eng.load(b, a)
    for t in range(ticks):
        dxdy = np.zeros((A, 1, 2), np.float32); act = np.zeros((A, 1), np.int32)
        for a in range(A):
            dd, _ = policy(11 + a, t, 1, False, 6)
            dxdy[a] = dd
        eng.set_actions(dxdy, act); eng.step()
        for a in range(A):
            oras[a].take_actions(dxdy[a], act[a]); oras[a].step()
            d = blob.diff(oras[a].dump(), eng.dump(a))
            assert d is None, "tick %d, %d cells: %s" % (t, sizes[a], d)
    fl = eng.flags()
    eng.close()
    assert not fl.any(), fl


def test_dense_clumps_of_many_cells_emulated(emu_lib, oracle_lib):
    from agarcl_amd import _capi
    run_clumps(lambda A, **cfg: _capi.BatchedEngine(A, lib=emu_lib, **cfg), oracle_lib)


@pytest.mark.gpu
def test_dense_clumps_of_many_cells_hip(hip_engine_cls, oracle_lib):
    run_clumps(hip_engine_cls, oracle_lib, ticks=40)


@pytest.mark.gpu
def test_short_square_root_is_sqrtf_on_every_float(hip_engine_cls):
    """The pair visit's 9-instruction square root (agar_core.inl ag_sqrtf_lean) equals the compiler's correctly rounded sqrtf bit for bit on
    all 2^32 bit patterns (zeros, denormals, the < 2^-96 range that falls back to the long form, infinities, NaNs)."""
    eng = hip_engine_cls(1, arena_size=100, num_pellets=10, num_viruses=0)
    assert eng.sqrt_check() == (0, 2 ** 64 - 1)
    eng.close()
