"""Observation kernels alone at 4096 arenas (mode 6 after 40 steps): us per call of the screen rasteriser (plain / agent view, band
rasteriser vs the pixel-wise cross-check kernel), the ram observation and the grid observation."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(3)
for t in range(40):
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32)); eng.step()
eng.sync()
def timeit(name, fn, n=20):
    fn(); eng.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    eng.sync(); print("%-44s %9.1f us" % (name, (time.perf_counter() - t0) / n * 1e6), flush=True)
scr = torch.empty((A, 84, 84, 3), dtype=torch.uint8, device='cuda'); av = torch.empty((A, 84, 84, 4), dtype=torch.uint8, device='cuda')
ram = torch.empty((A, 1, 152), dtype=torch.float32, device='cuda'); grid = torch.empty((A, 8, 128, 128), dtype=torch.int32, device='cuda')
for pw in ("0", "1"):
    os.environ["AGARCL_SCREEN_PIXELWISE"] = pw
    timeit("screen 84x84x3  %s" % ("pixel-wise" if pw == "1" else "band"), lambda: eng.screen_obs(84, 84, out_ptr=scr.data_ptr()))
    timeit("screen 84x84x4 agent view  %s" % ("pixel-wise" if pw == "1" else "band"), lambda: eng.screen_obs(84, 84, out_ptr=av.data_ptr(), agent_view=True), n=5)
os.environ["AGARCL_SCREEN_PIXELWISE"] = "0"
timeit("ram obs (16, 16, 8, 16)", lambda: eng.ram_obs(16, 16, 8, 16, out_ptr=ram.data_ptr()))
timeit("grid obs 8x128x128 persistent", lambda: eng.grid_obs(128, out_ptr=grid.data_ptr(), persistent=True))
