"""Where a SHORT timed window (bench.py --steps 20 --warmup 5, the driver's command) loses time against the steady state:
  (a) wall time of the 20-step window by the way the host waits for the end of it;  (b) GPU time of each of the first steps after a reset
(one HIP-event pair per step; the engine runs on torch's current stream)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd.vec_env import VecEnvironment as BatchedAgarEnv
import bench
A = 4096
cfg = dict(bench.CFG)
dev = torch.device("cuda:0")
def fresh():
    env = BatchedAgarEnv(A, device=0, strict_flags=False, **cfg); env.seed(np.arange(10000, 10000 + A, dtype=np.uint32)); env.reset(reset_ids=True); return env
g = torch.Generator(device=dev); g.manual_seed(1234)
N = 200
dxdy = (torch.rand((N, A, 1, 2), generator=g, device=dev) * 2 - 1).contiguous(); act = torch.zeros((N, A, 1), dtype=torch.int32, device=dev)
dp = [dxdy[k].data_ptr() for k in range(N)]; ap = [act[k].data_ptr() for k in range(N)]
def window(env, W, K, wait):
    eng = env.engine
    for k in range(W): eng.step_actions(dp[k], ap[k], 4)
    eng.timer_mark(0); eng.timer_mark(1); torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.timer_mark(0)
    for k in range(W, W + K): eng.step_actions(dp[k], ap[k], 4)
    eng.timer_mark(1)
    if wait == "spin":
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
    elif wait == "engsync": eng.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return el / K * 1e6, eng.timer_elapsed_ms() / K * 1e3
for wait in ("torch", "spin", "engsync", "torch", "spin", "engsync"):
    r = []
    for rep in range(5):
        env = fresh(); r.append(window(env, 5, 20, wait)); env.close()
    print("wait=%-8s  20-step window after 5 warm-ups: wall us/step %s | HIP-event us/step %s" % (wait, " ".join("%.2f" % a for a, _ in r), " ".join("%.2f" % b for _, b in r)))
for W in (5, 25, 100):
    r = []
    for rep in range(3):
        env = fresh(); r.append(window(env, W, 20, "torch")); env.close()
    print("warm-up %3d: wall %s | events %s" % (W, " ".join("%.2f" % a for a, _ in r), " ".join("%.2f" % b for _, b in r)))
# per-step GPU time of the first 40 steps after a reset
env = fresh(); eng = env.engine; evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
torch.cuda.synchronize(); evs[0].record()
for k in range(40): eng.step_actions(dp[k], ap[k], 4); evs[k + 1].record()
torch.cuda.synchronize()
print("per-step us (events between steps; includes the event record itself):", " ".join("%.1f" % (evs[k].elapsed_time(evs[k + 1]) * 1e3) for k in range(40)))
