"""Writes a few screen-observation frames (HIP rasteriser) as PNG files under gpurun_out/ for eyeballing."""
import sys; sys.path.insert(0, '.')
import numpy as np
from PIL import Image
from agarcl_amd import _capi
for name, cfg, steps in (("m6", dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6), 120), ("bots", dict(num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0), 200)):
    eng = _capi.BatchedEngine(2, **cfg); eng.seed(None, 5); eng.reset(reset_ids=True)
    rng = np.random.RandomState(0)
    for t in range(steps):
        eng.set_actions(rng.uniform(-1, 1, (2, 1, 2)).astype(np.float32), rng.randint(0, 3, (2, 1)).astype(np.int32)); eng.step()
    f = eng.screen_obs(336, 336)[0, 0][::-1]   # flip: PNG rows are top-down
    Image.fromarray(f).save('gpurun_out/screen_%s.png' % name)
    print(name, 'mass', eng.masses()[0], 'counts', eng.counts()[0])
