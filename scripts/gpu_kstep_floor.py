"""Floor of the general engine: single-cell arenas forced through k_step (AGARCL_NO_FRONT=1, and the in-kernel quiet run disabled by a
mass that... no: quiet_run still applies inside k_step).  Variants: viruses 0 / 25, action none / random."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import torch
from agarcl_amd.vec_env import VecEnvironment
A = 4096
for nv, rand_act, mass in ((0, False, 25), (25, False, 25), (25, True, 25), (0, True, 60), (25, True, 60)):
    env = VecEnvironment(A, num_viruses=nv, mode_number=0, strict_flags=False)
    env.seed(base_seed=5); env.reset(reset_ids=True)
    if mass != 25:
        from oracle import blob
        d = blob.parse(env.engine.dump(0)); d["players"][0]["cell_mass"][0] = mass; bb = blob.build(d)
        for a in range(A): env.engine.load(bb, a)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    dx = (torch.rand((64, A, 1, 2), generator=g, device=env.device) * 2 - 1)
    ac = torch.randint(0, 3, (64, A, 1), generator=g, device=env.device, dtype=torch.int32) if rand_act else torch.zeros((64, A, 1), dtype=torch.int32, device=env.device)
    for k in range(300): env.take_actions(dx[k % 64], ac[k % 64]); env.step()
    torch.cuda.synchronize(); t0 = time.time(); K = 400
    for k in range(K): env.take_actions(dx[k % 64], ac[k % 64]); env.step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / K * 1e6
    c = env.engine.counts().mean(axis=0)
    print('viruses %2d, actions %-6s start mass %3d: %.1f us/step; cells %.2f foods %.1f' % (nv, 'random' if rand_act else 'none', mass, dt, c[3], c[2]))
    env.close()
