"""Diagnostic (profile build): shader-clock cycles spent by the front kernel per arena (state load, quiet ticks)
next to the wall time per step of back-to-back launches, for a cell that stands still (no scans, no eats)."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('build_variants/lib_PROF.so'))
lib.agarcl_debug_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
def run(A, move, ticks, K=300):
    eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=1, lib=lib)  # mode 1: no regen, no decay
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) * move, np.zeros((A, 1), np.int32))
    for k in range(30): eng.step(ticks)
    eng.sync()
    out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    t0 = time.time()
    for k in range(K): eng.step(ticks)
    eng.sync(); wall = (time.time() - t0) / K * 1e6
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    per = out.astype(np.float64) / (A * K)
    print('A=%d move=%.1f ticks=%3d: load %.0f cyc, ticks %.0f cyc -> %.0f cyc/tick; wall %.2f us/step' % (A, move, ticks, per[0], per[1], per[1] / ticks, wall), flush=True)
    eng.close()
for A in (1024, 4096, 16384):
    for ticks in (1, 4, 16, 64):
        run(A, 0.0, ticks)
