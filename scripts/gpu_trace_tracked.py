"""Debug aid: the tracked-pellet test arenas on the HIP engine, state words per tick."""
import sys, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from agarcl_amd import _capi
from oracle import orabind, blob
orabind.build()
import test_tracked_pellet as T
A = len(T.CASES)
eng = _capi.BatchedEngine(A, **T.CFG)
oras = [orabind.OraEnv(**T.CFG) for _ in range(A)]
eng.seed(np.arange(40, 40 + A, dtype=np.uint32)); eng.reset(reset_ids=True)
for a, (o, (_, cell, pel, _)) in enumerate(zip(oras, T.CASES)):
    o.seed(40 + a); o.reset(True); b = T.arena_with(o, cell, pel); o.load(b); eng.load(b, a)
dxdy = np.array([[c[3]] for c in T.CASES], dtype=np.float32).reshape(A, 1, 2); act = np.zeros((A, 1), np.int32)
prev = None
for t in range(40):
    eng.set_actions(dxdy, act); eng.step()
    ar, pl = eng.arena_words(0); w = pl[0]
    cur = (int(w[9]), int(w[19]), int(w[22]), float(np.int32(ar[47]).view(np.float32)), float(np.int32(w[17]).view(np.float32)), float(np.int32(w[20]).view(np.float32)))
    if cur != prev: print(t, "eaten %d passes %d cidx %d slack %.3f safe_x %.2f cand_x %.2f" % cur, "fused", int(eng.L.agarcl_debug_fused(eng.h)), "work", eng.work()); prev = cur
