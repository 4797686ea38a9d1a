"""The headline workload (C2, 4096 arenas) with the boundary's HOST buffers: actions copied host -> device every step (agarcl_set_actions with host
pointers: 12 B per arena), rewards and dones copied back (agarcl_get_rewards / _dones: 9 B per arena), against the device-resident loop that
`value` is quoted on.  python scripts/gpu_pcie_rate.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
A = 4096
cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
eng = _capi.BatchedEngine(A, **cfg); eng.seed(None, 42); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(8)]
ac = [np.zeros((A, 1), np.int32) for _ in range(8)]
g = torch.Generator(device='cuda'); g.manual_seed(7)
dxdy = (torch.rand((8, A, 1, 2), generator=g, device='cuda') * 2 - 1).contiguous(); act = torch.zeros((8, A, 1), device='cuda', dtype=torch.int32)
def loop(K, host_in, host_out):
    for k in range(K):
        if host_in: eng.set_actions(mv[k % 8], ac[k % 8])
        else: eng.set_actions_device(dxdy[k % 8].data_ptr(), act[k % 8].data_ptr())
        eng.step()
        if host_out: r = eng.rewards(); d = eng.dones()
    eng.sync()
for name, hi, ho in (("device-resident actions, results left in HBM", False, False), ("host actions in, results left in HBM", True, False), ("host actions in, rewards + dones copied out every step", True, True)):
    loop(100, hi, ho); t0 = time.perf_counter(); K = 1000; loop(K, hi, ho); us = (time.perf_counter() - t0) / K * 1e6
    print("%-58s %.1f us per step = %.3g env-steps/s" % (name, us, A * 4 / us * 1e6), flush=True)
