"""Diagnostic (profile build): 100 MHz realtime stamps of every arena's front-part start / end in the LAST step of a
C2-like run (fresh random direction every step), with the number of pellet passes it made: who are the stragglers?"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('build_variants/lib_PROF.so'))
lib.agarcl_debug_prof_raw.argtypes = [C.c_void_p, C.c_void_p]
A = 4096
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0, lib=lib)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
zero = np.zeros((A, 1), np.int32)
acc = []
for k in range(80):
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), zero); eng.step(); 
    if k >= 40 and k % 30 != 29:
        eng.sync()
        raw = np.zeros((A, 16), np.uint64); lib.agarcl_debug_prof_raw(eng.h, raw.ctypes.data)
        st, en, ns = raw[:, 4].astype(np.int64), raw[:, 5].astype(np.int64), raw[:, 6].astype(np.int64)
        t0 = st.min(); d = (en - st) / 100.0
        acc.append((d, ns, (en.max() - t0) / 100.0, (st - t0) / 100.0))
d = np.concatenate([a[0] for a in acc]); ns = np.concatenate([a[1] for a in acc]); span = np.array([a[2] for a in acc]); sts = np.concatenate([a[3] for a in acc])
print('span (first start -> last end of ticks) per step: mean %.2f us, p90 %.2f' % (span.mean(), np.percentile(span, 90)))
print('start offsets: p50 %.2f p99 %.2f max %.2f us' % (np.percentile(sts, 50), np.percentile(sts, 99), sts.max()))
for n in range(0, 5):
    m = ns == n
    if m.any(): print('arenas with %d pellet passes: %.3f%% of arena-steps, in-kernel time p50 %.2f p90 %.2f max %.2f us' % (n, 100 * m.mean(), np.percentile(d[m], 50), np.percentile(d[m], 90), d[m].max()))
print('all: p50 %.2f p90 %.2f p99 %.2f max %.2f' % (np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max()))
