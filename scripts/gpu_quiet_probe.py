"""One configuration of the C2 workload for kernel-level timing under rocprofv3 (scripts/quiet_probe.sh).
usage: gpu_quiet_probe.py <move scale> <ticks per step> <arenas> [random|fixed]"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
from agarcl_amd import _capi
move, ticks, A = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else 'random'
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
acts = [(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) * move) for _ in range(16)]
zero = np.zeros((A, 1), np.int32)
for k in range(300):
    if mode == 'random' or k == 0: eng.set_actions(acts[k % 16], zero)
    eng.step(ticks)
eng.sync()
print('counts', eng.counts().mean(axis=0))
eng.close()
