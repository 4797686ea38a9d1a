"""Where does a mid-game step go?  The same warm state (4096 arenas, agents grown to mass 150, 400 steps of play on the product build) is
stepped 12 times by builds with one phase of the tick compiled out (build_variants/lib_ABL_<X>.so); the time difference to the full
build is that phase's share.  Results of the ablated builds are wrong by construction: timing only."""
import ctypes as C, os, sys, time
sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi, snapshot
A = 4096
cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
mode = sys.argv[1] if len(sys.argv) > 1 else "mid"
if mode == "m6": cfg["mode"] = 6
eng = _capi.BatchedEngine(A, **cfg)
eng.seed(None, 10000); eng.reset(reset_ids=True)
if mode == "mid":
    scfg = dict(num_agents=1, ticks_per_step=4, arena_size=1000, num_bots=0, reward_type=1, c_death=0, mode_number=0, pellet_regen=True)
    sn = snapshot.save_arena(eng, 0, scfg)
    for pl in sn["players"]:
        for cell in pl["cells"]: cell["mass"] = 150
    for a in range(A):
        sn["seed"] = 10000 + a; snapshot.load_arena(eng, a, sn, reset_ids=True)
rng = np.random.RandomState(1)
mv = [rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32) for _ in range(16)]
ac = [rng.randint(0, 3, size=(A, 1)).astype(np.int32) for _ in range(16)]
for k in range(400): eng.set_actions(mv[k % 16], ac[k % 16]); eng.step()
eng.sync()
blobs = [eng.dump(a) for a in range(A)]
print("warm state: mean counts", eng.counts().mean(axis=0), flush=True)
eng.close()
import torch
dmv = [torch.as_tensor(m, device='cuda') for m in mv]; dac = [torch.as_tensor(a_, device='cuda') for a_ in ac]
for name in sys.argv[2:]:
    TICKS = 4
    if "@" in name: name, t_ = name.split("@"); TICKS = int(t_)
    lib = _capi.bind(C.CDLL(os.path.join('build_variants', 'lib_%s.so' % name))) if name != "product" else None
    e2 = _capi.BatchedEngine(A, lib=lib, **cfg) if lib else _capi.BatchedEngine(A, **cfg)
    os.environ["AGARCL_FUSED"] = "0"
    for a in range(A): e2.load(blobs[a], a)
    ts = []
    for rep in range(3):
        for a in range(0, A, 1): pass
        e2.sync(); e2.timer_mark(0)
        for k in range(12): e2.step_actions(dmv[k % 16].data_ptr(), dac[k % 16].data_ptr(), TICKS)
        e2.timer_mark(1); ts.append(e2.timer_elapsed_ms() / 12 * 1e3)
        for a in range(A): e2.load(blobs[a], a)     # back to the warm state
    print("%-16s ticks %d %8.1f us/step (min of 3: %.1f)" % (name, TICKS, float(np.median(ts)), min(ts)), flush=True)
    e2.close()
