import sys, time
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
def run(A, move, K=600, ticks=4):
    eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) * move
    eng.set_actions(dxdy, np.zeros((A, 1), np.int32))
    for _ in range(30): eng.step(ticks)
    eng.sync()
    ts = []
    for k in range(K):
        t0 = time.perf_counter(); eng.step(ticks); eng.sync(); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts)
    # batches of 15 launches back to back for a throughput number
    t0 = time.perf_counter()
    for k in range(K): eng.step(ticks)
    eng.sync(); thr = (time.perf_counter() - t0) / K * 1e6
    print('A=%d move=%.1f: back-to-back %.1f us/launch; synced per-launch median %.1f p10 %.1f p90 %.1f max %.1f' % (A, move, thr, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), ts.max()), flush=True)
    eng.close()
for A in (4096,):
    run(A, 1.0); run(A, 0.0)
