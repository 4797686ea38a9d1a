"""Does hipDeviceScheduleSpin shorten the closing synchronisation of a short timed region?  (gpu_driver20.py's loop, both settings)"""
import ctypes, sys, time, os
sys.path.insert(0, '.')
mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
hip = ctypes.CDLL("libamdhip64.so")
if mode != "auto":
    flag = {"spin": 1, "yield": 2, "block": 4}[mode]
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(flag)))
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
res = []
for rep in range(6):
    env.seed(base_seed=10000); env.reset(reset_ids=True)
    for k in range(5): eng.step_actions(dp[k], ap[k], 4)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for k in range(5, 25): eng.step_actions(dp[k], ap[k], 4)
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6 / 20, e0.elapsed_time(e1) * 1e3 / 20))
print(mode, " | ".join("enq %.0f sync %.0f => %.2f us/step (events %.2f)" % r for r in res[1:]))
env.close()
