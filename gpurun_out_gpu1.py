import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import orabind, blob
from agarcl_amd import _capi
from lockstep import run_batched_lockstep
orabind.build()
cfgs = [
 (dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0), 300, 4),
 (dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6), 800, 8),
 (dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6), 800, 8),
 (dict(arena_size=300, num_pellets=300, num_viruses=5, mode=1), 300, 8),
]
A = 8
for cfg, steps, sticky in cfgs:
    eng = _capi.BatchedEngine(A, **cfg)
    oras = [orabind.OraEnv(**cfg) for _ in range(A)]
    t0 = time.time()
    ok, msg = run_batched_lockstep(eng, oras, steps, seeds=list(range(11, 11 + A)), sticky=sticky)
    print(cfg, ok, msg, '%.1fs' % (time.time() - t0), flush=True)
    if not ok:
        ok2, msg2 = run_batched_lockstep(eng, oras, steps, seeds=list(range(11, 11 + A)), sticky=sticky, rtol=1e-4)
        print('   with rtol 1e-4:', ok2, msg2, flush=True)
# quick timing
for A in (4096, 16384):
    eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    dxdy = rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32); act = np.zeros((A, 1), np.int32)
    eng.set_actions(dxdy, act)
    for _ in range(20): eng.step(4)
    eng.sync()
    t0 = time.time()
    K = 200
    for _ in range(K): eng.step(4)
    eng.sync()
    dt = time.time() - t0
    print('A=%d: %.1f us/step(4 ticks), %.3g arena-ticks/s' % (A, dt / K * 1e6, A * 4 * K / dt), flush=True)
    print('flags any', eng.flags().any(), 'mass mean', eng.masses().mean())
