"""k_screen_obs alone for several builds of the library on ONE box (box-to-box spread is larger than the differences looked for): every build in a
child process (AGARCL_HIP_SO), us per 4096 frames on the task-like states.   python scripts/gpu_screen_ab.py LIB [LIB ...]   (LIB = path or build_variants name)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from agarcl_amd.vec_env import VecEnvironment
A = 4096
for name, cfg in (("task3", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=3)), ("task1", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=1)),
                  ("task6", dict(arena_size=350, num_pellets=500, num_viruses=0, mode_number=6)), ("C3m6", dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode_number=6))):
    env = VecEnvironment(A, strict_flags=False, **cfg); env.seed(base_seed=10000); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(60):
        env.take_actions(torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1, torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32)); env.step()
    row = []
    for (W, av) in ((128, True), (84, False)):
        for _ in range(3): env.screen_obs(W, W, agent_view=av)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): env.screen_obs(W, W, agent_view=av)
        torch.cuda.synchronize()
        row.append("%%dx%%dx%%d %%6.1f" %% (W, W, 4 if av else 3, (time.perf_counter() - t0) / 30 * 1e6))
    print("  %%-6s %%s" %% (name, "   ".join(row)), flush=True)
    env.close()
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        path = lib if os.path.exists(lib) else os.path.join(ROOT, "build_variants", "lib_%s.so" % lib)
        env = dict(os.environ, AGARCL_HIP_SO=os.path.abspath(path))
        print("== %s (pass %d)" % (lib, rep), flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env)
