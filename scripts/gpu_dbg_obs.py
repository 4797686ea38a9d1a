import sys
sys.path.insert(0, '.')
import numpy as np, torch
from agarcl_amd import _capi
cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
A = 48
eng = _capi.BatchedEngine(A, **cfg)
eng.seed(None, 321); eng.reset(reset_ids=True)
rng = np.random.RandomState(2)
keep = torch.full((A, 1, 8, 128, 128), 7, dtype=torch.int32, device="cuda")
fresh = torch.empty((A, 1, 8, 128, 128), dtype=torch.int32, device="cuda")
for t in range(60):
    eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(A, 1)).astype(np.int32))
    eng.step()
    eng.grid_obs(128, out_ptr=keep.data_ptr(), persistent=True)
    eng.grid_obs(128, out_ptr=fresh.data_ptr())
    d = (keep != fresh)
    if d.any():
        idx = d.nonzero()
        print("t", t, "mismatches", int(d.sum()), "first", idx[0].tolist(), "keep", int(keep[tuple(idx[0])]), "fresh", int(fresh[tuple(idx[0])]), "channels", sorted(set(idx[:, 2].tolist())), "arenas", len(set(idx[:, 0].tolist())))
        break
else:
    print("no mismatch")
