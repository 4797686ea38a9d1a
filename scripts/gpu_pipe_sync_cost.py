"""What ordering a sub-batch's stream against the caller's stream costs (agarcl_pipe_fork / _join = hipEventRecord + hipStreamWaitEvent), and
what the RL surface gets from sub-batches once that is paid: AgarioVectorEnv mode 6 / C1, sub_batches 1 vs 2, full-batch step() vs the recv /
send halves with the policy running on each sub-batch's own stream (no cross-stream event at all)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import torch
from agarcl_amd.vector_env import AgarioVectorEnv
from agarcl_amd import _capi
dev = torch.device("cuda", 0)
N = 4096
def t_loop(fn, K=200, W=20):
    for _ in range(W): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    host = time.perf_counter() - t0; torch.cuda.synchronize()
    return host / K * 1e6, (time.perf_counter() - t0) / K * 1e6
# (a) bare fork + join on an idle pipe
pipe = _capi.PipelinedEngine(N, 2, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
cur = torch.cuda.current_stream().cuda_stream
print("fork+join, nothing enqueued: host %.1f us, total %.1f us per round" % t_loop(lambda: (pipe.fork(cur), pipe.join(cur))))
x = torch.zeros(1024, device=dev)
print("fork+join around a tiny torch op: host %.1f us, total %.1f us" % t_loop(lambda: (x.add_(1), pipe.fork(cur), pipe.join(cur))))
pipe.close()
for name, kw in (("mode6", dict(mode=6, num_viruses=25)), ("C1", dict(arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, dt=1.0 / 60)), ("C2", {})):
    for obs in ("none", "screen"):
        for k in (1, 2):
            venv = AgarioVectorEnv(N, obs_type=obs, sub_batches=k, strict_flags=False, number_steps=100000, **kw)
            venv.reset(seed=10000)
            g = torch.Generator(device=dev); g.manual_seed(0)
            def policy(n):   # a stand-in for inference: fresh action tensors produced on the CURRENT stream every step
                return torch.rand((n, 2), generator=g, device=dev) * 2 - 1, torch.randint(0, 3, (n,), generator=g, device=dev, dtype=torch.int32)
            h, t = t_loop(lambda: venv.step(policy(N)), K=100, W=20)
            line = "%-6s obs=%-6s sub_batches=%d  step(): host %.1f us, %.1f us per vector step" % (name, obs, k, h, t)
            if k > 1:   # the halves, each range's policy on the range's own stream: no cross-stream ordering at all
                streams = [p.torch_stream() for p in venv._parts]
                def halves():
                    for j in range(k):
                        with torch.cuda.stream(streams[j]):
                            venv.recv(j); venv.send(policy(venv.ranges[j][1]), j)
                h, t = t_loop(halves, K=100, W=20)
                line += "   | halves on own streams: host %.1f us, %.1f us" % (h, t)
            print(line, flush=True)
            venv.close()
