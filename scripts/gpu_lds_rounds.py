"""Is a several-player k_step launch a second round short of LDS?  us per step (no observation) of the paper's task 7 (agent + 1 bot) by arena count and capacities.
python scripts/gpu_lds_rounds.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agarcl_amd.vec_env import VecEnvironment
def run(A, tag, **kw):
    cfg = dict(arena_size=350, num_pellets=500, num_viruses=0, num_bots=1, mode_number=7); cfg.update(kw)
    env = VecEnvironment(A, strict_flags=False, **cfg); env.seed(base_seed=10000); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    acts = [(torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32)) for _ in range(16)]
    for t in range(100): env.take_actions(*acts[t % 16]); env.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(200): env.take_actions(*acts[t % 16]); env.step()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 200 * 1e6
    print("%-40s A=%5d  %7.1f us per step  %6.1f M env-steps/s   flags %s" % (tag, A, us, A * 4 / us, hex(env.engine.poll_flags()) if hasattr(env.engine, "poll_flags") else "?"), flush=True)
    env.close()
for A in (3072, 3584, 4096):
    run(A, "default capacities")
run(4096, "cap_viruses=8", cap_viruses=8)
run(4096, "cap_viruses=8 cap_foods=64", cap_viruses=8, cap_foods=64)
run(4096, "mode 3 + 1 bot, cap_viruses=8 cap_foods=64", mode_number=3, cap_viruses=8, cap_foods=64)
