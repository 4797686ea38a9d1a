"""A/B timing of alternative builds on the C3 / mode 6 loop (diagnostic). usage: gpu_ab6.py libA.so libB.so ..."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
def run(path, A=4096, K=150, W=250, ticks=4):
    lib = _capi.bind(C.CDLL(path))
    eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6, lib=lib)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    g = torch.Generator(device='cuda'); g.manual_seed(1234)
    dxdy = (torch.rand((K + W, A, 1, 2), generator=g, device='cuda') * 2 - 1).contiguous(); act = torch.randint(0, 3, (K + W, A, 1), generator=g, device='cuda', dtype=torch.int32)
    torch.cuda.synchronize()
    for k in range(W): eng.set_actions_device(dxdy[k].data_ptr(), act[k].data_ptr()); eng.step(ticks)
    eng.sync(); t0 = time.perf_counter()
    for k in range(W, W + K): eng.set_actions_device(dxdy[k].data_ptr(), act[k].data_ptr()); eng.step(ticks)
    eng.sync(); us = (time.perf_counter() - t0) / K * 1e6
    print('%-40s %.1f us/step  -> %.3e env-steps/s   (mean cells %.2f, mass sum %d)' % (path.split('/')[-1], us, A * ticks / us * 1e6, eng.counts()[:, 3].mean(), int(eng.masses().sum())), flush=True)
    eng.close()
for p in sys.argv[1:]:
    run(p)
