"""What does the slowest mode-6 arena cost when it has a SIMD to itself, and when it shares it with three copies of itself?
PROF + LEVELS build (build_variants/lib_PROFL.so: -DAGAR_PROFILE -DAGAR_PROFILE_LEVELS).  Steps 4096 mode-6 arenas 300 times, takes
the slowest arena of the next step (scripts/gpu_arena_spread.py: the same arenas are slow step after step), copies its state through the
JSON snapshot into EVERY arena of engines of 1024 / 2048 / 4096 arenas (1 / 2 / 4 wavefronts per SIMD), drives them with one action for
all and prints cycles per arena-step, visited levels per tick and cycles per visited level."""
import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi, snapshot
os.environ.setdefault("AGARCL_NO_FRONT", "1")
lib = _capi.bind(C.CDLL(os.environ.get('PROF_SO', 'build_variants/lib_PROFL.so')))
CFG = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
SCFG = dict(num_agents=1, ticks_per_step=4, arena_size=1000, num_bots=0, reward_type=1, c_death=0, mode_number=6, pellet_regen=True)
A = 4096
eng = _capi.BatchedEngine(A, lib=lib, **CFG)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
acts = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(8)]
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
for k in range(300): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(4)
eng.sync()
out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
raw = np.zeros((A, 16), np.uint64)
eng.set_actions(mv[4], acts[4]); eng.step(4); eng.sync()
lib.agarcl_debug_prof_raw(eng.h, raw.ctypes.data)
raw[:, 13:16] = 0; raw[raw > 10**9] = 0
tot = raw.astype(np.float64).sum(axis=1)
order = np.argsort(-tot)
print("4096 mixed arenas: arena-step cycles mean %.0f max %.0f; slowest arenas %s" % (tot.mean(), tot.max(), order[:4].tolist()))
picks = {"slowest": int(order[0]), "median": int(order[A // 2])}
snaps = {k: snapshot.save_arena(eng, a, SCFG) for k, a in picks.items()}
cells = eng.counts()[:, 3]
for k, a in picks.items(): print("   %s arena %d: %d cells, %.0f cycles in the mixed launch" % (k, a, cells[a], tot[a]))
eng.close()
for name, sn in snaps.items():
    for A2 in (1024, 2048, 4096):
        e2 = _capi.BatchedEngine(A2, lib=lib, **CFG)
        e2.seed(None, 5); e2.reset(reset_ids=True)
        for a in range(A2): snapshot.load_arena(e2, a, sn, reset_ids=True)
        m1 = np.tile(np.array([[[0.3, -0.2]]], np.float32), (A2, 1, 1)); a1 = np.zeros((A2, 1), np.int32)
        e2.sync(); lib.agarcl_debug_prof(e2.h, out.ctypes.data, 1)
        K = 5
        t0 = time.time()
        for k in range(K): e2.set_actions(m1, a1); e2.step(4)
        e2.sync(); wall = (time.time() - t0) / K * 1e6
        lib.agarcl_debug_prof(e2.h, out.ctypes.data, 1)
        per = out.astype(np.float64) / (A2 * K); per[per > 1e9] = 0   # (phases 4 / 5 of this build carry a wrapped difference)
        lv, lh, passes = per[13] / 4.0, per[14] / 4.0, per[15] / 4.0; per[13:16] = 0   # (LEVELS build: slots 13..15 are counts, not cycles)
        print("%s x %d arenas (%d waves / SIMD): %.0f cycles per arena-step (relaxation+move %.0f), wall %.0f us / step; per tick: %.1f levels visited, %.1f with a touching pair, %.1f touch passes; %.0f cycles per visited level"
              % (name, A2, A2 // 1024, per.sum(), per[3], wall, lv, lh, passes, per[3] / 4 / max(lv, 1e-9)))
        e2.close()
