"""Does it matter WHICH arenas share a SIMD?  k_step runs one wavefront per arena, four wavefronts per SIMD, and workgroup b lands on SIMD
b mod 1024 (scripts/microbench/wg_placement.hip).  Mode 6 at 4096 arenas: per-arena cycle counts of one step are taken with the PROF build,
then the SAME 4096 arena states are laid out in three orders on the product build and timed:
  identity   arena a in slot a (what the engine does)
  snake      by cost: the 1024 heaviest on 1024 different SIMDs, the next 1024 in reverse order, ... (every SIMD gets a similar sum)
  clustered  the four heaviest on SIMD 0, the next four on SIMD 1, ... (the worst case)
usage: gpu_placement.py [steps]     needs build_variants/lib_PROF.so (python -m agarcl_amd.build --profile)"""
import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
os.environ.setdefault("AGARCL_NO_FRONT", "1")
A = 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
prof = _capi.bind(C.CDLL(os.environ.get('PROF_SO', 'build_variants/lib_PROF.so')))
e0 = _capi.BatchedEngine(A, lib=prof, **cfg)
e0.seed(None, 10000); e0.reset(reset_ids=True)
rng = np.random.RandomState(0)
acts = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(8)]
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
for k in range(300): e0.set_actions(mv[k % 8], acts[k % 8]); e0.step(4)
e0.sync()
out = np.zeros(16, np.uint64); prof.agarcl_debug_prof(e0.h, out.ctypes.data, 1)
raw = np.zeros((A, 16), np.uint64)
e0.set_actions(mv[0], acts[0]); e0.step(4); e0.sync()
prof.agarcl_debug_prof_raw(e0.h, raw.ctypes.data)
cost = raw.astype(np.float64).sum(axis=1)
print("per-arena cycles of one step: mean %.0f  p90 %.0f  p99 %.0f  max %.0f" % (cost.mean(), np.percentile(cost, 90), np.percentile(cost, 99), cost.max()), flush=True)
blobs = [e0.dump(a) for a in range(A)]
e0.close()
rank = np.argsort(-cost)                     # rank[r] = arena with the r-th highest cost
s = np.arange(1024)
snake = np.empty(A, np.int64); clustered = np.empty(A, np.int64)
for k in range(4):
    snake[1024 * k + s] = rank[1024 * k + (1023 - s if k & 1 else s)]
    clustered[1024 * k + s] = rank[4 * s + k]
orders = {"identity": np.arange(A), "snake": snake, "clustered": clustered}
lib = _capi.hip_lib()
engs = {}
for name, order in orders.items():
    e = _capi.BatchedEngine(A, **cfg)
    e.seed(None, 10000); e.reset(reset_ids=True)
    for slot in range(A): e.load(blobs[order[slot]], slot)
    simd_sum = np.array([cost[order[np.arange(4) * 1024 + j]].sum() for j in range(1024)])
    print("%-10s per-SIMD cycle sums: mean %.0f max %.0f ; heaviest arena shares its SIMD with costs %s" % (name, simd_sum.mean(), simd_sum.max(),
          [int(cost[order[(int(np.nonzero(order == rank[0])[0][0]) % 1024) + 1024 * k]]) for k in range(4)]), flush=True)
    engs[name] = (e, order)
res = {n: [] for n in orders}
for rep in range(3):
    for name, (e, order) in engs.items():
        e.set_actions(mv[0][order], acts[0][order]); e.step(4); e.sync()
        e.set_actions(mv[rep % 8][order], acts[rep % 8][order])      # (one action set per repetition: no host copy inside the timed steps)
        e.timer_mark(0)
        for k in range(steps): e.step(4)
        e.timer_mark(1)
        res[name].append(e.timer_elapsed_ms() / steps * 1e3)
for name in orders: print("%-10s us per step: %s" % (name, ["%.1f" % x for x in res[name]]))
