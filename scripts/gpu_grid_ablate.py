"""Which part of k_grid_obs costs what?  Wall time per call (persistent tensor, 4096 mode-6 arenas stepped between calls is NOT done here:
the same state is observed again and again, which keeps the undo list at its steady size) with channels switched off."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
for k in range(100):
    eng.set_actions(rng.uniform(-1, 1, (A, 1, 2)).astype(np.float32), rng.randint(0, 3, (A, 1)).astype(np.int32)); eng.step()
for name, kw in (("all channels", {}), ("no pellets", dict(pellets=False)), ("no viruses", dict(viruses=False)), ("no cells", dict(cells=False, others=False)), ("pellets only", dict(cells=False, others=False, viruses=False)), ("mask only", dict(cells=False, others=False, viruses=False, pellets=False))):
    ch = 1 + (2 if kw.get('pellets', True) else 0) + (2 if kw.get('viruses', True) else 0) + (1 if kw.get('cells', True) else 0) + (2 if kw.get('others', True) else 0)
    out = torch.zeros((A, 1, ch, 128, 128), dtype=torch.int32, device="cuda")
    for k in range(20): eng.grid_obs(128, out_ptr=out.data_ptr(), persistent=True, **kw)
    eng.sync(); t0 = time.time()
    K = 300
    for k in range(K): eng.grid_obs(128, out_ptr=out.data_ptr(), persistent=True, **kw)
    eng.sync(); print("%-14s %.1f us per call (wall, %d calls)" % (name, (time.time() - t0) / K * 1e6, K))
    del out
