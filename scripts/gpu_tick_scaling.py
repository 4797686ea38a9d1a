import sys, time
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
def run(A, ticks, K=200, **cfg):
    eng = _capi.BatchedEngine(A, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = np.zeros((A, 1), np.int32)
    eng.set_actions(dxdy, act)
    for _ in range(20): eng.step(ticks)
    eng.sync(); t0 = time.time()
    for _ in range(K): eng.step(ticks)
    eng.sync(); dt = (time.time() - t0) / K
    eng.close()
    return dt * 1e6
C2 = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
for A in (256, 1024, 4096, 8192, 16384):
    r = [run(A, t, **C2) for t in (1, 4, 16)]
    print('A=%5d us/launch ticks=1:%.1f 4:%.1f 16:%.1f  -> per-tick %.2f us, fixed %.1f us' % (A, r[0], r[1], r[2], (r[2]-r[1])/12, r[1] - 4*(r[2]-r[1])/12), flush=True)
C3 = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
for A in (4096,):
    r = [run(A, t, **C3) for t in (1, 4, 16)]
    print('mode6 A=%5d us/launch ticks=1:%.1f 4:%.1f 16:%.1f' % (A, r[0], r[1], r[2]), flush=True)
