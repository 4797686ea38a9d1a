"""Per-phase shader-clock breakdown of k_step in the mid-game regime (mode 0, agents grown to mass 150: ~2.4 cells, ~7 foods)."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
from oracle import blob
lib = _capi.bind(C.CDLL('build_variants/lib_PROF.so'))
names = ['load', 'tick_pre', 'pl_load/bots', 'selfcol', 'virus', 'pellets', 'stats/food', 'emit/split/add', 'recomb/decay/store', 'regen/end', 'env_post', 'store', 'kinematics', 'remove', 'sort', 'plcol/foods']
A, K = 4096, 200
eng = _capi.BatchedEngine(A, lib=lib, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
eng.seed(None, 900); eng.reset(reset_ids=True)
d = blob.parse(eng.dump(0)); d["players"][0]["cell_mass"][0] = int(sys.argv[1]) if len(sys.argv) > 1 else 150; bb = blob.build(d)
for a in range(A): eng.load(bb, a)
rng = np.random.RandomState(1)
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(16)]
ac = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(16)]
for k in range(400): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
eng.sync()
out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
for k in range(K): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
eng.sync(); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
per = out.astype(np.float64) / (A * K)
print('cycles per wave per launch (4 ticks): total %.0f; counts (pellets, viruses, foods, cells) %s' % (per.sum(), eng.counts().mean(axis=0)))
for n, v in zip(names, per): print('   %-22s %8.0f  %5.1f%%' % (n, v, 100 * v / per.sum()))
