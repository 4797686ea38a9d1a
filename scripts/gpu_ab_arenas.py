"""A/B of library builds across batch sizes on the C2 loop (diagnostic). usage: gpu_ab_arenas.py lib.so ... """
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
def run(path, A, K=150, W=40, ticks=4):
    lib = _capi.bind(C.CDLL(path))
    eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0, lib=lib)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    g = torch.Generator(device='cuda'); g.manual_seed(1234)
    dxdy = (torch.rand((16, A, 1, 2), generator=g, device='cuda') * 2 - 1).contiguous(); act = torch.zeros((A, 1), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    for k in range(W): eng.set_actions_device(dxdy[k % 16].data_ptr(), act.data_ptr()); eng.step(ticks)
    eng.sync(); t0 = time.perf_counter()
    for k in range(K): eng.set_actions_device(dxdy[k % 16].data_ptr(), act.data_ptr()); eng.step(ticks)
    eng.sync(); us = (time.perf_counter() - t0) / K * 1e6
    print('%-28s A=%6d  %.2f us/step  -> %.3e env-steps/s' % (path.split('/')[-1], A, us, A * ticks / us * 1e6), flush=True)
    eng.close()
for A in (4096, 16384, 65536, 262144):
    for p in sys.argv[1:]:
        run(p, A)
