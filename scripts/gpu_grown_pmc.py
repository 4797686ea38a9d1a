"""Mid-game workload (mode 0, agents grown to mass argv[1], random actions, every arena-step through k_step) as a stand-alone command
for rocprofv3 --pmc runs."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
from oracle import blob
A, K = 4096, 200
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
eng.seed(None, 900); eng.reset(reset_ids=True)
d = blob.parse(eng.dump(0)); d["players"][0]["cell_mass"][0] = int(sys.argv[1]) if len(sys.argv) > 1 else 60; bb = blob.build(d)
for a in range(A): eng.load(bb, a)
rng = np.random.RandomState(1)
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(16)]
ac = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(16)]
for k in range(300): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
eng.sync(); t0 = time.time()
for k in range(K): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
eng.sync(); print("%.1f us/step (incl. host action upload)" % ((time.time() - t0) / K * 1e6), eng.counts().mean(axis=0))
