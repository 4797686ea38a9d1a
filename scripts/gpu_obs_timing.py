"""Timing of the grid-observation kernel with parts switched off (diagnostic): where does its time go?"""
import sys, time, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from agarcl_amd import _capi
A = 4096
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
for t in range(100):
    eng.set_actions(rng.uniform(-1, 1, (A, 1, 2)).astype(np.float32), rng.randint(0, 3, (A, 1)).astype(np.int32)); eng.step()
out = torch.empty((A, 8, 128, 128), dtype=torch.int32, device='cuda')
for flags in ((True, True, True, True),) if len(sys.argv) > 1 else ((True, True, True, True), (False, False, False, True), (False, False, True, False), (True, False, False, False), (False, False, False, False)):
    for _ in range(3): eng.grid_obs(128, *flags, out_ptr=out.data_ptr())
    eng.sync(); t0 = time.perf_counter()
    for _ in range(20): eng.grid_obs(128, *flags, out_ptr=out.data_ptr())
    eng.sync(); us = (time.perf_counter() - t0) / 20 * 1e6
    C = 1 + flags[0] + 2 * flags[1] + 2 * flags[2] + 2 * flags[3]
    print('cells=%d others=%d viruses=%d pellets=%d: %.0f us, %d channels -> %.2f TB/s' % (*flags, us, C, A * C * 128 * 128 * 4 / us / 1e6))
