"""How many visited levels of the self-collision relaxation could be taken two at a time?  ("tried in round 4")
Builds the kernel source as a host emulation with -DAGAR_STATS_LEVELS (counters inside self_collisions: a level-time and the next one are
mergeable iff the pairs of both that touch BEFORE the visit share no cell; a misspeculation is a pair of the second level-time that starts to
touch only after the first moved) and runs mode 6 on it.  usage: cpu_level_stats.py [arenas] [steps]      CPU only."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = "/tmp/libagarcl_emu_stats.so"
subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O2", "-DAGAR_CPU_EMU", "-DAGAR_STATS_LEVELS", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                       "-Wno-unknown-pragmas", "-Wno-unused-function", "-o", SO, os.path.join(ROOT, "agarcl_amd", "csrc", "agar_engine.hip"), "-lm"])
from agarcl_amd import _capi
raw = ctypes.CDLL(SO); lib = _capi.bind(raw)
A = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = _capi.BatchedEngine(A, lib=lib, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
eng.seed(None, 10000); eng.reset(None, True)
rng = np.random.RandomState(1)
for t in range(steps):
    if t == 100: raw.agarcl_emu_level_stats_reset()
    eng.set_actions(rng.uniform(-1, 1, (A, 1, 2)).astype(np.float32), rng.randint(0, 3, (A, 1)).astype(np.int32)); eng.step()
out = (ctypes.c_long * (8 + 33 * 3))(); raw.agarcl_emu_level_stats(out); v = list(out)
print(dict(zip(["calls", "visited", "merged_away", "misspec", "calls_dense", "visited_dense", "merged_dense", "misspec_dense"], v[:8])))
print("cells: visited levels, merged away, misspeculated, share merged")
for n in range(33):
    a, b, c = v[8 + 3 * n: 11 + 3 * n]
    if a: print("%5d: %8d %8d %6d  %.2f" % (n, a, b, c, b / a))
