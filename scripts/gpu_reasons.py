"""Why does the lean front part hand an arena-step to the general engine?  (-DAGAR_PROFILE_REASONS build, C2 workload)"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('agarcl_amd/libagarcl_hip_why.so'))
names = ['not single-cell/food-free', 'eject/split possible', 'virus in reach', 'virus regeneration', 'generator exhausted at regen', 'anti-team bookkeeping', 'several pellets in reach', 'growth reaches a 2nd pellet']
for A, steps in ((4096, 3000), (65536, 400)):
    eng = _capi.BatchedEngine(A, lib=lib, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(16)]
    act = np.zeros((A, 1), np.int32)
    for k in range(steps): eng.set_actions(mv[k % 16], act); eng.step()
    eng.sync()
    q = np.zeros(16, np.int32); lib.agarcl_debug_qstat(eng.h, q.ctypes.data)
    tot = q[4:12].sum()
    print('A=%d, %d steps: unfinished arena-steps %d (%.2e of all)' % (A, steps, q[0], q[0] / (A * steps)))
    for n, v in zip(names, q[4:12]): print('   %-32s %7d  %5.1f%%' % (n, v, 100.0 * v / max(tot, 1)))
    eng.close()
