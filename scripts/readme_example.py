import sys; sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
env = VecEnvironment(4096, num_agents=1, arena_size=1000, num_pellets=1000, mode_number=0)
env.seed(base_seed=10000); env.reset()
dxdy = torch.rand((4096, 1, 2), device='cuda') * 2 - 1; act = torch.zeros((4096, 1), dtype=torch.int32, device='cuda')
for _ in range(50):
    env.take_actions(dxdy, act); env.step()
env.sync()
print(env.rewards.shape, env.dones().shape, env.masses.float().mean().item(), env.packed[env.engine.last_slot()].shape)
from agarcl_amd import agarcl
e = agarcl.GridEnvironment(1, 4, 1000, True, 1000, 25, 0, 1, 0, 6)
e.configure_observation({"grid_size": 32}); e.seed(1); e.reset(); e.take_actions([(0.1, 0.2, 0)]); print(e.step(), e.get_state()[0].shape)
