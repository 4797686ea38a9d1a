"""Where the driver-style run (`bench.py --steps 20 --warmup 5`) spends its time: per-step HIP-event times of the first 40 steps after a
reset, host enqueue time of the 20 timed steps, and the closing synchronisation.  Run on the GPU box."""
import sys, time
sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
import bench

A = 4096
cfg = dict(bench.CFG)
for rep in range(3):
    env = VecEnvironment(A, strict_flags=False, **cfg)
    env.seed(base_seed=10000); env.reset(reset_ids=True)
    eng = env.engine
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    N = 45
    dx = (torch.rand((N, A, 1, 2), generator=g, device=env.device) * 2 - 1).contiguous()
    ac = torch.zeros((N, A, 1), dtype=torch.int32, device=env.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    for k in range(5):
        eng.set_actions_device(dx[k].data_ptr(), ac[k].data_ptr()); eng.step(4)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(5, 25):
        eng.set_actions_device(dx[k].data_ptr(), ac[k].data_ptr()); eng.step(4)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("rep %d: enqueue 20 steps %.1f us, first sync %.1f us, second sync %.1f us, total/20 = %.2f us/step, events/20 = %.2f us/step" % (
        rep, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6 / 20, e0.elapsed_time(e1) * 1e3 / 20))
    # per-step event times of further steps (events between steps add their own cost; shows the trend only)
    ev[0].record()
    for k in range(25, 45):
        eng.set_actions_device(dx[k].data_ptr(), ac[k].data_ptr()); eng.step(4); ev[k - 24].record()
    torch.cuda.synchronize()
    print("   per-step (with an event after every step):", " ".join("%.1f" % (ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(20)))
    env.close()
