"""Floor of the C2 step (default 4096 arenas; argv[1] = arena count): the same step with (almost) no pellets -- no pellet passes, no eats -- against the real workload."""
import sys, time
sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for npel in (1000, 1000, 2, 2):
    env = VecEnvironment(A, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=npel, num_viruses=0, mode_number=0, strict_flags=False)
    env.seed(base_seed=10000); env.reset(reset_ids=True)
    eng = env.engine
    g = torch.Generator(device=env.device); g.manual_seed(1)
    dx = (torch.rand((64, A, 1, 2), generator=g, device=env.device) * 2 - 1).contiguous()
    ac = torch.zeros((64, A, 1), dtype=torch.int32, device=env.device)
    dp = [dx[k].data_ptr() for k in range(64)]; ap = [ac[k].data_ptr() for k in range(64)]
    for k in range(200): eng.step_actions(dp[k % 64], ap[k % 64], 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 2000 if A <= 16384 else 400
    for k in range(K): eng.step_actions(dp[k % 64], ap[k % 64], 4)
    e1.record(); torch.cuda.synchronize()
    w = eng.work()
    print("pellets %4d: %.2f us/step; passes/step %.1f" % (npel, e0.elapsed_time(e1) * 1e3 / K, w[2] / (K + 200)))
    env.close()
