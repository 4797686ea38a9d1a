"""Per-arena cost spread of the general engine (PROF build): a launch lasts as long as its slowest arena.  Prints, for mode 6 at 4096 arenas,
the cycles one arena-step takes by cell count (mean / max), and the slowest arenas' phase split.   PROF_SO=build_variants/lib_PROF.so"""
import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
os.environ.setdefault("AGARCL_NO_FRONT", "1")
lib = _capi.bind(C.CDLL(os.environ.get('PROF_SO', 'build_variants/lib_PROF.so')))
names = ['load', 'tick_pre', 'pl_load', 'selfcol', 'virus', 'pellets', 'stats/food', 'emit/split', 'recomb/decay', 'regen/end', 'env_post', 'store', 'move', 'remove', 'sort', 'plcol/foods']
A = 4096
MID = len(sys.argv) > 1 and sys.argv[1] == "mid"
eng = _capi.BatchedEngine(A, lib=lib, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0 if MID else 6)
eng.seed(None, 10000); eng.reset(reset_ids=True)
if MID:   # agents grown to mass 150 (bench.py --workload mid)
    from agarcl_amd import snapshot
    scfg = dict(num_agents=1, ticks_per_step=4, arena_size=1000, num_bots=0, reward_type=1, c_death=0, mode_number=0, pellet_regen=True)
    sn = snapshot.save_arena(eng, 0, scfg)
    for pl in sn["players"]:
        for cell in pl["cells"]: cell["mass"] = 150
    for a in range(A):
        sn["seed"] = 10000 + a; snapshot.load_arena(eng, a, sn, reset_ids=True)
rng = np.random.RandomState(0)
acts = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(8)]
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
for k in range(400 if MID else 300): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(4)
eng.sync()
out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
raw = np.zeros((A, 16), np.uint64)
for trial in range(3):
    k = (400 if MID else 300) + trial
    cells_before = eng.counts()[:, 3].copy()
    eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(4); eng.sync()
    lib.agarcl_debug_prof_raw(eng.h, raw.ctypes.data)
    tot = raw.astype(np.float64).sum(axis=1)
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    print("step %d: arena-step cycles: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f   (max / mean %.2f)" % (k, tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.max() / tot.mean()))
    print("   mean phase split: " + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, raw.astype(np.float64).mean(axis=0))))
    med = np.argsort(tot)[A // 2 - 2:A // 2 + 2]
    for a in med: print("   median arena %4d (cells %d): %.0f = " % (a, cells_before[a], tot[a]) + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, raw[a]) if v > 0.03 * tot[a]))
    for n in range(1, 17):
        m = cells_before == n
        if m.sum(): print("   cells %2d: %4d arenas  mean %8.0f  max %8.0f" % (n, m.sum(), tot[m].mean(), tot[m].max()))
    for a in np.argsort(-tot)[:5]:
        print("   slowest arena %4d (cells %d): %.0f = " % (a, cells_before[a], tot[a]) + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, raw[a]) if v > 0.03 * tot[a]))
