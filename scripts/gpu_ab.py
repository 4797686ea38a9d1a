import sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from agarcl_amd import _capi
from oracle import orabind
from lockstep import run_batched_lockstep
orabind.build()
def run(lib, A, ticks, K=200, **cfg):
    eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), np.zeros((A, 1), np.int32))
    for _ in range(20): eng.step(ticks)
    eng.sync(); t0 = time.time()
    for _ in range(K): eng.step(ticks)
    eng.sync(); dt = (time.time() - t0) / K
    eng.close(); return dt * 1e6
C2 = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
C3 = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
for name in sys.argv[1:]:
    lib = _capi.bind(C.CDLL(name))
    cfg = dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6)
    eng = _capi.BatchedEngine(8, lib=lib, **cfg); oras = [orabind.OraEnv(**cfg) for _ in range(8)]
    ok, msg = run_batched_lockstep(eng, oras, 300, seeds=np.arange(11, 19), sticky=8, every=10)
    print(name, 'parity', ok, msg, flush=True)
    for A in (256, 4096, 16384):
        r = [run(lib, A, t, **C2) for t in (1, 4, 16)]
        print('  C2 A=%5d us/launch ticks=1:%.1f 4:%.1f 16:%.1f -> per-tick %.2f fixed %.1f' % (A, r[0], r[1], r[2], (r[2]-r[1])/12, r[1]-4*(r[2]-r[1])/12), flush=True)
    print('  mode6 A=4096 4 ticks: %.1f us' % run(lib, 4096, 4, **C3), flush=True)
