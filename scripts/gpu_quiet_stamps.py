"""Diagnostic (profile build): 100 MHz realtime stamps of every arena's front-kernel start / end in the LAST step:
spread of starts, per-arena duration, and overall span, to see where k_quiet's duration goes."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('agarcl_amd/libagarcl_hip_prof.so'))
lib.agarcl_debug_prof_raw.argtypes = [C.c_void_p, C.c_void_p]
def run(A, move, ticks, mode):
    eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=mode, lib=lib)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) * move, np.zeros((A, 1), np.int32))
    for k in range(37): eng.step(ticks)
    eng.sync()
    raw = np.zeros((A, 16), np.uint64); lib.agarcl_debug_prof_raw(eng.h, raw.ctypes.data)
    st, en = raw[:, 4].astype(np.int64), raw[:, 5].astype(np.int64)
    ok = st > 0
    st, en = st[ok], en[ok]; t0 = st.min()
    d = (en - st) * 10.0  # ns
    print('A=%d move=%.1f ticks=%d mode=%d: span %.2f us; start spread p50 %.2f p99 %.2f max %.2f us; per-arena in-kernel time p50 %.2f p90 %.2f max %.2f us'
          % (A, move, ticks, mode, (en.max() - t0) / 100.0, np.percentile(st - t0, 50) / 100.0, np.percentile(st - t0, 99) / 100.0, (st.max() - t0) / 100.0,
             np.percentile(d, 50) / 1000, np.percentile(d, 90) / 1000, d.max() / 1000), flush=True)
    eng.close()
for mode in (1, 0):
    for move in (0.0, 1.0):
        for ticks in (1, 4):
            run(4096, move, ticks, mode)
