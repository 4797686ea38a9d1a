"""Per-arena cost spread of the C1 population (agent + 4 bot kinds; PROF build): cycles of one arena-step by total cell count, the slowest arenas'
phase split, over several steps (bot ticks fall on every 10th tick: a 4-tick step holds one in 2 of 5 steps).   python scripts/gpu_arena_spread_c1.py"""
import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL(os.environ.get('PROF_SO', 'build_variants/lib_PROF.so')))
names = ['load', 'tick_pre', 'pl_load/bot', 'selfcol/move', 'virus', 'pellets', 'stats/food', 'emit/split', 'recomb/decay', 'regen/end', 'env_post', 'store', 'kinematics', 'remove(+simple)', 'sort', 'plcol/foods']
A = 4096
eng = _capi.BatchedEngine(A, lib=lib, num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
acts = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(8)]
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for k in range(W): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(4)
eng.sync()
out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
raw = np.zeros((A, 16), np.uint64)
for trial in range(5):
    k = W + trial
    cells_before = eng.counts()[:, 3].copy(); foods = eng.counts()[:, 2].copy()
    eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(4); eng.sync()
    lib.agarcl_debug_prof_raw(eng.h, raw.ctypes.data)
    tot = raw.astype(np.float64).sum(axis=1)
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    print("step %d (ticks %d..%d): arena-step cycles: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f   (max / mean %.2f)" % (k, 4 * k, 4 * k + 3, tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.max() / tot.mean()))
    print("   mean phase split: " + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, raw.astype(np.float64).mean(axis=0))))
    for lo, hi in ((5, 5), (6, 7), (8, 11), (12, 19), (20, 200)):
        m = (cells_before >= lo) & (cells_before <= hi)
        if m.sum(): print("   cells %3d-%3d: %4d arenas  mean %8.0f  max %8.0f   (with foods: %d)" % (lo, hi, m.sum(), tot[m].mean(), tot[m].max(), (foods[m] > 0).sum()))
    for a in np.argsort(-tot)[:4]:
        print("   slowest arena %4d (cells %d, foods %d): %.0f = " % (a, cells_before[a], foods[a], tot[a]) + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, raw[a]) if v > 0.03 * tot[a]))
