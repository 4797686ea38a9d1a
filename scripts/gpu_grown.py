"""Mid-game regime: mode-0 arenas whose agents have grown (so that eject / split fire and foods lie around): how long does a step
take once every arena needs the general engine?  (random actions ~ U{0,1,2}, 25 viruses, 4096 arenas)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
from oracle import blob, orabind
cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for mass in (30, 60, 150, 400):
    eng = _capi.BatchedEngine(A, **cfg)
    eng.seed(None, 900); eng.reset(reset_ids=True)
    b0 = eng.dump(0); d = blob.parse(b0); d["players"][0]["cell_mass"][0] = mass; bb = blob.build(d)
    for a in range(A):   # same start state everywhere is fine for timing: the actions differ
        eng.load(bb, a)
    rng = np.random.RandomState(1)
    mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(16)]
    ac = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(16)]
    for k in range(400): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
    eng.sync(); w0 = eng.work(reset=True)
    t0 = time.time(); K = 300
    for k in range(K): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
    eng.sync(); dt = (time.time() - t0) / K * 1e6
    w = eng.work(); c = eng.counts().mean(axis=0)
    print('start mass %4d: %.1f us/step (incl. ~%d us of host action upload), general arena-steps per step %.0f of %d, mean cells %.2f foods %.1f, fused=%d' % (mass, dt, 0, w[1] / K, A, c[3], c[2], eng.L.agarcl_debug_fused(eng.h)))
    eng.close()
