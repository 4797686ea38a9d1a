import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agarcl_amd import _capi
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
import ctypes
lib = _capi.bind(ctypes.CDLL(os.environ['AGAR_LIB'])) if os.environ.get('AGAR_LIB') else None
eng = _capi.BatchedEngine(A, lib=lib, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
eng.seed(None, 10000); eng.reset(reset_ids=True)
rng = np.random.RandomState(0)
eng.set_actions(rng.uniform(-1, 1, size=(A, 1, 2)).astype(np.float32), np.zeros((A, 1), np.int32))
for _ in range(30): eng.step(ticks)
eng.sync()
