"""Throughput of the C1 population (agent + 4 bot kinds, 250x250, 500 pellets, 10 viruses, mode 0) on the GPU, and of a
3-agent mode-6 arena (diagnostic / DESIGN.md table)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
def run(name, A, K=100, W=150, **cfg):
    na = cfg.get('num_agents', 1)
    eng = _capi.BatchedEngine(A, **cfg); eng.seed(None, 42); eng.reset(reset_ids=True)
    g = torch.Generator(device='cuda'); g.manual_seed(7)
    dxdy = (torch.rand((16, A, na, 2), generator=g, device='cuda') * 2 - 1).contiguous(); act = torch.randint(0, 3, (16, A, na), generator=g, device='cuda', dtype=torch.int32)
    for k in range(W): eng.set_actions_device(dxdy[k % 16].data_ptr(), act[k % 16].data_ptr()); eng.step()
    eng.sync(); t0 = time.perf_counter()
    for k in range(K): eng.set_actions_device(dxdy[k % 16].data_ptr(), act[k % 16].data_ptr()); eng.step()
    eng.sync(); us = (time.perf_counter() - t0) / K * 1e6
    print('%s: A=%d  %.1f us/step -> %.3e env-steps/s (mean cells %.1f, flags %d)' % (name, A, us, A * 4 / us * 1e6, eng.counts()[:, 3].mean(), int((eng.flags() != 0).sum())), flush=True)
    eng.close()
run('C1 population', 4096, num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0)
run('3 agents mode 6', 4096, num_agents=3, arena_size=250, num_pellets=500, num_viruses=10, mode=6)
run('mode 8 (agent + bot)', 4096, num_agents=1, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=8)
