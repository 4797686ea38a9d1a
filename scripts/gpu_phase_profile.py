import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('build_variants/lib_PROF.so'))
lib.agarcl_debug_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
names = ['load', 'tick_pre', 'pl_load', 'selfcol', 'virus', 'pellets', 'stats/food', 'emit/split/add', 'recomb/decay/store', 'regen/end', 'env_post', 'store', 'kinematics', 'remove', 'sort', 'plcol/foods']
def run(A, K=100, ticks=4, rand_act=False, **cfg):
    eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), np.zeros((A, 1), np.int32))
    acts = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) if rand_act else np.zeros((A, 1), np.int32) for _ in range(8)]
    mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
    for k in range(200 if rand_act else 10): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(ticks)
    eng.sync()
    out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    t0 = time.time()
    for k in range(K): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(ticks)
    eng.sync(); wall = (time.time() - t0) / K * 1e6
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    per = out.astype(np.float64) / (A * K)
    print('A=%d cfg=%s: cycles per wave per launch (%d ticks): total %.0f, wall %.1f us/launch' % (A, cfg.get('mode'), ticks, per.sum(), wall))
    for n, v in zip(names, per): print('   %-22s %8.0f' % (n, v))
    print('   mean counts (pellets, viruses, foods, cells):', eng.counts().mean(axis=0))
C2 = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
run(4096, **C2)
run(4096, rand_act=True, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
run(4096, rand_act=True, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
