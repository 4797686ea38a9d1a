"""Fixed cost and per-tick cost of the single-launch step at 4096 arenas: step time against ticks per step (2 pellets: no pellet work)."""
import sys
sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
A = 4096
for npel in (2, 1000):
    env = VecEnvironment(A, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=npel, num_viruses=0, mode_number=0, strict_flags=False)
    env.seed(base_seed=10000); env.reset(reset_ids=True)
    eng = env.engine
    g = torch.Generator(device=env.device); g.manual_seed(1)
    dx = (torch.rand((64, A, 1, 2), generator=g, device=env.device) * 2 - 1).contiguous()
    ac = torch.zeros((64, A, 1), dtype=torch.int32, device=env.device)
    dp = [dx[k].data_ptr() for k in range(64)]; ap = [ac[k].data_ptr() for k in range(64)]
    out = []
    for ticks in (1, 2, 4, 8, 16):
        for k in range(100): eng.step_actions(dp[k % 64], ap[k % 64], ticks)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K = 1000
        for k in range(K): eng.step_actions(dp[k % 64], ap[k % 64], ticks)
        e1.record(); torch.cuda.synchronize()
        out.append("%d ticks: %.2f" % (ticks, e0.elapsed_time(e1) * 1e3 / K))
    print("pellets %4d, us/step:" % npel, " | ".join(out))
    env.close()
