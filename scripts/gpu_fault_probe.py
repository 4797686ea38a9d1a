"""Do builds of the library agree arena by arena?  N steps of mode 6 at a given arena count for several builds / launch pins; every run is a
child process (a queue fault aborts the process, a hang is cut after 150 s); per-arena digests (counts, masses, rewards) are compared with
the FIRST variant's and the arenas that differ are listed.
usage: gpu_fault_probe.py <arenas> <steps> VARIANT[:ENV=VAL...] ...   (VARIANT = product | build_variants/lib_<VARIANT>.so)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from agarcl_amd import _capi
A, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
eng = _capi.BatchedEngine(A, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
eng.seed(None, 31000); eng.reset(reset_ids=True)
rng = np.random.RandomState(1)
for t in range(steps):
    eng.set_actions(rng.uniform(-1, 1, (A, 1, 2)).astype(np.float32), rng.randint(0, 3, (A, 1)).astype(np.int32)); eng.step()
eng.sync()
cn = eng.counts()
np.save(out, np.concatenate([cn, eng.masses().reshape(A, -1), eng.rewards().reshape(A, -1).astype(np.int64), eng.flags().reshape(A, 1).astype(np.int64)], axis=1))
print("ok", int(cn[:, 3].sum()), "cells")
''' % ROOT
arenas, steps = sys.argv[1], sys.argv[2]
ref = None
for var in sys.argv[3:]:
    lib, *pins = var.split(":")
    env = dict(os.environ)
    if lib != "product": env["AGARCL_HIP_SO"] = os.path.join(ROOT, "build_variants", "lib_%s.so" % lib)
    for kv in pins: k, v = kv.split("="); env[k] = v
    out = os.path.join(tempfile.mkdtemp(), "d.npy")
    try:
        p = subprocess.run([sys.executable, "-c", CHILD, arenas, steps, out], env=env, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        print("%-28s HUNG (150 s)" % var, flush=True); continue
    msg = ""
    if p.returncode == 0:
        d = np.load(out)
        if ref is None: ref = d; msg = "(reference)"
        else:
            bad = np.nonzero((d != ref).any(axis=1))[0]
            msg = "equal to the reference in every arena" if len(bad) == 0 else "%d arenas differ: %s" % (len(bad), bad[:12].tolist())
    print("%-28s rc %d  %s  %s %s" % (var, p.returncode, p.stdout.strip()[-40:], msg, " | ".join(l for l in p.stderr.splitlines() if "HSA" in l or "fault" in l.lower())[:200]), flush=True)
