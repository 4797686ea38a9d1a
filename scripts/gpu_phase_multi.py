"""Per-phase cycle sums of the diagnostic build (-DAGAR_PROFILE -> build_variants/lib_PROF.so) for arenas with several players: C1 (agent + 4 bot
kinds), bench/main.cpp's Tick/N populations.  python scripts/gpu_phase_multi.py"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import os
import numpy as np
from agarcl_amd import _capi
lib = _capi.bind(C.CDLL('build_variants/lib_PROF.so'))
lib.agarcl_debug_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
names = ['load', 'tick_pre', 'pl_load/bot', 'selfcol/move', 'virus', 'pellets', 'stats/food', 'emit/split/add', 'recomb/decay/store', 'regen/end', 'env_post', 'store', 'kinematics', 'remove', 'sort', 'plcol/foods']
def run(tag, A, K=60, ticks=4, tick_only=False, **cfg):
    eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    na = cfg.get('num_agents', 1)
    rng = np.random.RandomState(0)
    mv = [rng.uniform(-1, 1, size=(A, max(na, 1), 2)).astype(np.float32) for _ in range(8)]
    ac = [rng.randint(0, 3, size=(A, max(na, 1))).astype(np.int32) for _ in range(8)]
    def step(k):
        if tick_only: eng.tick(ticks)
        else: eng.set_actions(mv[k % 8], ac[k % 8]); eng.step(ticks)
    for k in range(40): step(k)
    eng.sync()
    out = np.zeros(16, np.uint64); lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    t0 = time.time()
    for k in range(K): step(k)
    eng.sync(); wall = (time.time() - t0) / K * 1e6
    lib.agarcl_debug_prof(eng.h, out.ctypes.data, 1)
    per = out.astype(np.float64) / (A * K)
    print('%s A=%d: cycles per wave per launch (%d ticks): total %.0f, wall %.1f us/launch (host-paced)' % (tag, A, ticks, per.sum(), wall))
    for n, v in zip(names, per): print('   %-22s %8.0f  %5.1f%%' % (n, v, 100 * v / per.sum()))
    print('   mean counts (pellets, viruses, foods, cells):', eng.counts().mean(axis=0))
    eng.close()
if os.environ.get('PHASE_LIGHT'):   # round 6: light single-player arenas through the general engine (run with AGARCL_NO_FRONT=1), the paper's tasks 1, 3, 7
    run('task3 (mode 3)', 4096, K=100, arena_size=350, num_pellets=500, num_viruses=0, mode=3)
    run('task1 (mode 1)', 4096, K=100, arena_size=350, num_pellets=500, num_viruses=0, mode=1)
    run('task7 (mode 7 + bot)', 4096, K=100, arena_size=350, num_pellets=500, num_viruses=0, num_bots=1, mode=7)
    sys.exit(0)
if os.environ.get('PHASE_BIG'):   # round 6: the configurations scripts/gpu_config_sweep.py found slow
    run('normal + 25 bots', 4096, K=20, num_agents=1, arena_size=1000, num_pellets=1000, num_viruses=0, num_bots=25, mode=0)
    run('3 agents mode 6', 4096, K=20, num_agents=3, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
    run('normal + 4 bots', 4096, K=40, num_agents=1, arena_size=1000, num_pellets=1000, num_viruses=0, num_bots=4, mode=0)
    sys.exit(0)
run('C1', 4096, num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
run('Tick/10', 4096, tick_only=True, num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, dt=1.0 / 60, example_bots=10)
run('Tick/30', 4096, tick_only=True, num_agents=0, arena_size=250, num_pellets=500, num_viruses=10, mode=0, dt=1.0 / 60, example_bots=30)
run('C3m6', 4096, arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
