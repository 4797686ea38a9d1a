"""PMC target for k_screen_obs alone: 20 launches of one frame shape on a task-like state (AGARCL_SCR_ABL and AGARCL_HIP_SO select the
ablation / build).  argv: state (task3 | task1 | task6 | C3m6)  W  agent_view(0|1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agarcl_amd.vec_env import VecEnvironment
STATES = dict(task3=dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=3), task1=dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=1),
              task6=dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=6), C3m6=dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode_number=6))
name, W, av = sys.argv[1], int(sys.argv[2]), bool(int(sys.argv[3]))
A = 4096
env = VecEnvironment(A, strict_flags=False, **STATES[name]); env.seed(base_seed=10000); env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for t in range(60):
    env.take_actions(torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32)); env.step()
for _ in range(20):
    env.screen_obs(W, W, agent_view=av)
torch.cuda.synchronize()
env.close()
