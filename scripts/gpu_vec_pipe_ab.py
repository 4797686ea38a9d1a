"""AgarioVectorEnv.step() (full batch, a device policy between the steps: what a learner's sampling loop does) with sub_batches = 1 / 2 / 4 on the
workloads the default choice (vec_env.default_sub_batches) is about: us per vector step, host us per step.   python scripts/gpu_vec_pipe_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from agarcl_amd.vector_env import AgarioVectorEnv
A = 4096
CASES = (("mode 6, 1000x1000, 25 viruses, no obs", dict(obs_type="none", mode=6, num_viruses=25)),
         ("mode 6 + 84x84x3 screen", dict(obs_type="screen", screen_len=84, mode=6, num_viruses=25)),
         ("mode 6 + grid", dict(obs_type="grid", mode=6, num_viruses=25)),
         ("task 6 (350x350, 500 pellets, mode 6, 128x128x4 agent view)", dict(obs_type="screen", screen_len=128, agent_view=True, mode=6, arena_size=350, num_pellets=500, num_viruses=0)),
         ("task 10 (1 bot, mode 10, 128x128x4 agent view)", dict(obs_type="screen", screen_len=128, agent_view=True, mode=10, arena_size=350, num_pellets=500, num_viruses=0, num_bots=1)),
         ("C1-like (4 bots, 250x250, no obs)", dict(obs_type="none", mode=0, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4)),
         ("task 3 (quiet, 128x128x4 agent view)", dict(obs_type="screen", screen_len=128, agent_view=True, mode=3, arena_size=350, num_pellets=500, num_viruses=0)))
side = torch.cuda.Stream() if "--stream" in sys.argv else None      # --stream: the sampling loop on a stream of its own instead of the legacy default stream
if side is not None:
    torch.cuda.set_stream(side)
for name, kw in CASES[:2] + CASES[3:4] if "--short" in sys.argv else CASES:
    row = []
    for k in (1, 2, 4):
        venv = AgarioVectorEnv(A, sub_batches=k, strict_flags=False, number_steps=100000, **kw)
        venv.reset(seed=10000)
        dev = venv.device
        g = torch.Generator(device=dev); g.manual_seed(0)
        def policy():
            return torch.rand((A, 2), generator=g, device=dev) * 2 - 1, torch.randint(0, 3, (A,), generator=g, device=dev, dtype=torch.int32)
        for _ in range(40): venv.step(policy())
        torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        fixed = policy() if "--fixed" in sys.argv else None      # --fixed: one action batch for every step (no policy kernels between the steps)
        for _ in range(100):
            a = fixed if fixed is not None else policy(); h0 = time.perf_counter(); venv.step(a); host += time.perf_counter() - h0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        to = venv.pipe.pipe.spin_timeouts() if venv.pipe is not None else 0
        row.append("k=%d (%d concurrent%s): %6.1f us (host %5.1f)" % (k, venv.concurrent_sub_batches, ", SPIN TIME-OUT" if to else "", dt / 100 * 1e6, host / 100 * 1e6))
        venv.close()
    print("%-62s %s" % (name, "   ".join(row)), flush=True)
