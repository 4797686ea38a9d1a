"""Does a synchronous device-to-host copy right before a short timed region (bench.py resets the work counters there) slow the region?"""
import sys, time
sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
import bench
A = 4096
env = VecEnvironment(A, strict_flags=False, **dict(bench.CFG))
eng = env.engine
g = torch.Generator(device=env.device); g.manual_seed(1234)
dx = (torch.rand((25, A, 1, 2), generator=g, device=env.device) * 2 - 1).contiguous()
ac = torch.zeros((25, A, 1), dtype=torch.int32, device=env.device)
dp = [dx[k].data_ptr() for k in range(25)]; ap = [ac[k].data_ptr() for k in range(25)]
for variant in ("plain", "plain", "prewarmed_events", "plain"):
    res = []
    for rep in range(6):
        env.seed(base_seed=10000); env.reset(reset_ids=True)
        for k in range(5): eng.step_actions(dp[k], ap[k], 4)
        if variant == "work_reset_before": eng.work(reset=True)
        torch.cuda.synchronize(); torch.cuda.synchronize()
        if variant == "sleep_1ms_before": time.sleep(0.001)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if variant == "prewarmed_events": e0.record(); e1.record(); torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        for k in range(5, 25): eng.step_actions(dp[k], ap[k], 4)
        e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append(((t2 - t0) * 1e6 / 20, e0.elapsed_time(e1) * 1e3 / 20))
    print("%-20s" % variant, " | ".join("%.2f us/step (events %.2f)" % r for r in res))
env.close()
