"""The lane-parallel strip scan of players_collision against the lane-0 replay it gates (diagnostic build -DAG_PLCOL_CHECK ->
build_variants/lib_PLCHK.so: the replay runs whenever the necessary test fires and flag 0x4000 is raised when the scan's verdict differs from
the replay's result count).  python scripts/gpu_plcol_check.py"""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('build_variants/lib_PLCHK.so'))
def run(name, A, steps, **cfg):
    na = cfg.get('num_agents', 1)
    eng = _capi.BatchedEngine(A, lib=lib, **cfg); eng.seed(None, 77); eng.reset(reset_ids=True)
    rng = np.random.RandomState(1)
    mv = [rng.uniform(-1, 1, size=(A, max(na, 1), 2)).astype(np.float32) for _ in range(8)]
    ac = [rng.randint(0, 3, size=(A, max(na, 1))).astype(np.int32) for _ in range(8)]
    eaten0 = None
    for k in range(steps):
        if na: eng.set_actions(mv[k % 8], ac[k % 8]); eng.step(4)
        else: eng.tick(4)
    eng.sync()
    fl = eng.flags()
    print('%-40s A=%d steps=%d: arenas with a scan/replay difference %d, other flags %s, mean cells %.1f' % (name, A, steps, int(((fl & 0x4000) != 0).sum()),
          sorted(set(int(f) & ~0x4000 for f in fl if int(f) & ~0x4000)), eng.counts()[:, 3].mean()), flush=True)
    eng.close()
run('C1 (agent + 4 bot kinds)', 4096, 1500, num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
run('5 agents mode 6, 150x150', 2048, 600, num_agents=5, arena_size=150, num_pellets=300, num_viruses=0, mode=6)
run('3 agents + 6 bots, 120x120', 2048, 1000, num_agents=3, arena_size=120, num_pellets=300, num_viruses=4, num_bots=6, mode=0)
run('2 agents + 12 bots + 8 ExampleBots, 200x200', 1024, 1000, num_agents=2, arena_size=200, num_pellets=400, num_viruses=5, num_bots=12, example_bots=8, mode=0)
run('Tick/20 with 4 agents mode 6', 1024, 500, num_agents=4, arena_size=250, num_pellets=500, num_viruses=10, example_bots=20, mode=6, dt=1.0 / 60)
