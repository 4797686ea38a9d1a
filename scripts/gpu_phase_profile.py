import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('agarcl_amd/libagarcl_hip_prof.so'))
lib.agarcl_debug_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
names = ['load', 'tick_pre', 'pl_load', 'move', 'virus', 'pellets', 'stats/food', 'emit/split/add', 'recomb/decay/store', 'end_of_tick', 'env_post', 'store']
def run(A, K=100, ticks=4, **cfg):
    eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), np.zeros((A, 1), np.int32))
    for _ in range(10): eng.step(ticks)
    eng.sync()
    out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    t0 = time.time()
    for _ in range(K): eng.step(ticks)
    eng.sync(); wall = (time.time() - t0) / K * 1e6
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    per = out.astype(np.float64) / (A * K)
    print('A=%d cfg=%s: cycles per wave per launch (%d ticks): total %.0f, wall %.1f us/launch' % (A, cfg.get('mode'), ticks, per[:12].sum(), wall))
    for n, v in zip(names, per): print('   %-22s %8.0f' % (n, v))
C2 = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
run(256, **C2); run(4096, **C2)
run(4096, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
