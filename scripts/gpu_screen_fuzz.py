"""k_screen_obs (the band kernel) against k_screen_obs_pixelwise (every thread shades pixels, walking the whole list: AGARCL_SCREEN_PIXELWISE=1), byte for byte, over
many states, zooms and frame shapes: the band kernel's boxes, runs, marks and look-back pixels are all shortcuts the pixel-wise kernel does not take.
python scripts/gpu_screen_fuzz.py [seed] [arenas]   -> one line per configuration, "mismatching frames" must be 0 everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agarcl_amd.vec_env import VecEnvironment
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
A = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
CONFIGS = (dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=3), dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=1),
           dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=6), dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode_number=6),
           dict(arena_size=120, num_pellets=400, num_viruses=6, mode_number=6), dict(arena_size=250, num_pellets=250, num_viruses=10, mode_number=0, num_bots=4),
           dict(arena_size=80, num_pellets=64, num_viruses=25, mode_number=6), dict(arena_size=900, num_pellets=1200, num_viruses=40, mode_number=5))
SHAPES = ((128, 128), (84, 84), (96, 72), (64, 200), (256, 256), (37, 53), (8, 8), (512, 384), (130, 66))
bad_total = 0
for ci, cfg in enumerate(CONFIGS):
    env = VecEnvironment(A, strict_flags=False, **cfg); env.seed(base_seed=seed * 100000 + ci * 7919); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(seed * 31 + ci)
    n_agents = 1
    for phase, steps in enumerate((3, 40, 120)):
        for t in range(steps):
            env.take_actions(torch.rand((A, n_agents, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, n_agents), generator=g, device="cuda", dtype=torch.int32)); env.step()
        for (W, H) in SHAPES:
            for av in (True, False):
                os.environ.pop("AGARCL_SCREEN_PIXELWISE", None)
                f1 = env.screen_obs(W, H, agent_view=av).clone()
                os.environ["AGARCL_SCREEN_PIXELWISE"] = "1"
                f2 = env.screen_obs(W, H, agent_view=av).clone()
                os.environ.pop("AGARCL_SCREEN_PIXELWISE", None)
                bad = int((f1.reshape(A, -1) != f2.reshape(A, -1)).any(dim=1).sum())
                bad_total += bad
                if bad: print("MISMATCH cfg %d phase %d %dx%d av=%d: %d of %d frames" % (ci, phase, W, H, av, bad, A), flush=True)
    print("cfg %d %s: done, mismatching frames so far %d" % (ci, cfg, bad_total), flush=True)
    env.close()
print("screen fuzz seed %d: %d arenas x %d configs x 3 checkpoints x %d shapes x 2 views, mismatching frames: %d" % (seed, A, len(CONFIGS), len(SHAPES), bad_total))
sys.exit(1 if bad_total else 0)
