"""Per-phase shader-clock breakdown of the general engine (k_step) on mode 6 / C1 with the -DAGAR_PROFILE build."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
import os
os.environ.setdefault('AGARCL_NO_FRONT', '1')   # (the front kernel's profile stamps share slots 4 / 5 / 7)
lib = _capi.bind(C.CDLL(os.environ.get('PROF_SO', 'build_variants/lib_PROF.so')))
names = ['load', 'tick_pre', 'pl_load/bots', 'selfcol (relaxation)', 'virus', 'pellets', 'stats/food', 'emit/split/add', 'recomb/decay/store', 'regen/end', 'env_post', 'store', 'move', 'remove', 'sort', 'plcol/foods']
def run(A, K=100, ticks=4, **cfg):
    eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    na = cfg.get('num_agents', 1)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    acts = [rng.randint(0, 3, size=(A, na)).astype(np.int32) for _ in range(8)]
    mv = [rng.uniform(-1, 1, size=(A, na, 2)).astype(np.float32) for _ in range(8)]
    for k in range(200): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(ticks)
    eng.sync()
    out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    t0 = time.time()
    for k in range(K): eng.set_actions(mv[k % 8], acts[k % 8]); eng.step(ticks)
    eng.sync(); wall = (time.time() - t0) / K * 1e6
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    per = out.astype(np.float64) / (A * K)
    print('A=%d cfg=%s: cycles per wave per launch (%d ticks): total %.0f, wall %.1f us/launch (incl. host action upload)' % (A, cfg, ticks, per.sum(), wall))
    for n, v in zip(names, per): print('   %-24s %9.0f  %5.1f%%' % (n, v, 100 * v / per.sum()))
    print('   mean counts (pellets, viruses, foods, cells):', eng.counts().mean(axis=0))
    eng.close()
run(4096, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
run(4096, num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
