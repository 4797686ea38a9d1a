"""Phase breakdown of k_step on QUIET single-cell arenas (AGARCL_NO_FRONT=1): the general kernel's skeleton."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('build_variants/lib_PROF.so'))
names = ['load', 'tick_pre(+quiet run)', 'pl_load/bots', 'selfcol', 'virus', 'pellets', 'stats/food', 'emit/split/add', 'recomb/decay/store', 'regen/end', 'env_post', 'store', 'kinematics', 'remove', 'sort', 'plcol/foods']
A, K = 4096, 200
eng = _capi.BatchedEngine(A, lib=lib, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
eng.seed(None, 900); eng.reset(reset_ids=True)
rng = np.random.RandomState(1)
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(16)]
ac = np.zeros((A, 1), np.int32)
for k in range(100): eng.set_actions(mv[k % 16], ac); eng.step()
eng.sync()
out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
for k in range(K): eng.set_actions(mv[k % 16], ac); eng.step()
eng.sync(); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
per = out.astype(np.float64) / (A * K)
print('cycles per wave per launch (4 ticks): total %.0f' % per.sum())
for n, v in zip(names, per): print('   %-22s %8.0f  %5.1f%%' % (n, v, 100 * v / per.sum()))
