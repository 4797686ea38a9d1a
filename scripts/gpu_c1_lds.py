"""How much of a C1 step (agent + 4 bot kinds, 4096 arenas) is the LDS-limited residency of its wavefronts: the same population at several arena
counts (whole launches of 8 / 10 / 12 / 16 wavefronts per CU) and with smaller food capacities (less LDS per wavefront).  python scripts/gpu_c1_lds.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
C1 = dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0)
def run(name, A, K=150, W=150, **cfg):
    na = cfg.get('num_agents', 1)
    eng = _capi.BatchedEngine(A, **cfg); eng.seed(None, 42); eng.reset(reset_ids=True)
    g = torch.Generator(device='cuda'); g.manual_seed(7)
    dxdy = (torch.rand((16, A, na, 2), generator=g, device='cuda') * 2 - 1).contiguous(); act = torch.randint(0, 3, (16, A, na), generator=g, device='cuda', dtype=torch.int32)
    for k in range(W): eng.set_actions_device(dxdy[k % 16].data_ptr(), act[k % 16].data_ptr()); eng.step()
    eng.sync(); t0 = time.perf_counter()
    for k in range(K): eng.set_actions_device(dxdy[k % 16].data_ptr(), act[k % 16].data_ptr()); eng.step()
    eng.sync(); us = (time.perf_counter() - t0) / K * 1e6
    print('%-28s A=%5d  %.1f us/step -> %.3e env-steps/s, %.1f ns per arena-step (flags %d)' % (name, A, us, A / us * 1e6, us / A * 1e3, int((eng.flags() != 0).sum())), flush=True)
    eng.close()
for A in (1024, 2048, 2560, 3072, 4096, 8192): run('C1', A, **C1)
for fc in (64, 32, 16): run('C1 cap_foods=%d' % fc, 4096, cap_foods=fc, **C1)
run('C1 cap_foods=16 cap_viruses=16', 4096, cap_foods=16, cap_viruses=16, **C1)
