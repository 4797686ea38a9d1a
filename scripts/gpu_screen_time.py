"""k_screen_obs alone on task-like states: us per 4096 frames for the frame shapes in use.  python scripts/gpu_screen_time.py"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd.vec_env import VecEnvironment
A = 4096
for name, cfg in (("task3 (350x350, 500 pellets, mode 3)", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=3)),
                  ("task1 (squared pellets, mode 1)", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=1)),
                  ("task6 (350x350, 500 pellets, mode 6)", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=6)),
                  ("task7 (350x350, 500 pellets, 1 bot, mode 7)", dict(arena_size=350, num_pellets=500, num_viruses=0, num_bots=1, mode_number=7)),
                  ("C3m6 (1000x1000, 1000 pellets, 25 viruses, mode 6)", dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode_number=6))):
    env = VecEnvironment(A, strict_flags=False, **cfg); env.seed(base_seed=10000); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(60):
        env.take_actions(torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32)); env.step()
    for (W, H, av) in ((84, 84, False), (84, 84, True), (128, 128, False), (128, 128, True)):
        res = []
        for r05 in ("0",):
            os.environ["AGARCL_SCREEN_R05"] = r05
            out = env.screen_obs(W, H, agent_view=av)
            for _ in range(3): env.screen_obs(W, H, agent_view=av)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): env.screen_obs(W, H, agent_view=av)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 20 * 1e6)
        os.environ["AGARCL_SCREEN_R05"] = "0"
        print("%-52s %3dx%3dx%d: %7.1f us per %d frames" % (name, W, H, 4 if av else 3, res[0], A), flush=True)
    env.close()
