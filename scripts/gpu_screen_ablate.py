"""Which part of k_screen_obs costs what: the kernel alone on task-like states with parts switched off (measurement build:
python -m agarcl_amd.build --variant SCRABL -DAG_SCR_ABL; AGARCL_HIP_SO=build_variants/lib_SCRABL.so python scripts/gpu_screen_ablate.py).
AGARCL_SCR_ABL bits: 1 no entity list, 2 no painting, 4 no post-processing pass, 8 no global stores, 16 no background fill, 32 return at once, 64 / 128 return behind the first / second barrier."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
A = 4096
STATES = (("task3", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=3)),
          ("task1", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=1)),
          ("task6", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=6)),
          ("C3m6", dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode_number=6)))
ABL = (0, 2, 4, 7, 15, 31, 128, 64, 32)
for name, cfg in STATES:
    env = VecEnvironment(A, strict_flags=False, **cfg); env.seed(base_seed=10000); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(60):
        env.take_actions(torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32)); env.step()
    for (W, H, av) in ((128, 128, True), (84, 84, False)):
        row = []
        for ab in ABL:
            os.environ["AGARCL_SCR_ABL"] = str(ab)
            for _ in range(3): env.screen_obs(W, H, agent_view=av)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): env.screen_obs(W, H, agent_view=av)
            torch.cuda.synchronize()
            row.append("%d:%.1f" % (ab, (time.perf_counter() - t0) / 20 * 1e6))
        print("%-6s %3dx%3dx%d  us per %d frames by ablation bits  %s" % (name, W, H, 4 if av else 3, A, "  ".join(row)), flush=True)
    env.close()
