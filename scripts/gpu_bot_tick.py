"""What a bot-decision tick (every 10th) costs against an ordinary one: single-tick launches of 4096 arenas timed one by one (host clock around
launch + sync: ~10 us of floor in every figure), by tick index modulo 10.  python scripts/gpu_bot_tick.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
def run(name, A=4096, rounds=12, **cfg):
    eng = _capi.BatchedEngine(A, **cfg); eng.seed(None, 42); eng.reset(reset_ids=True)
    for _ in range(200): eng.tick(1)
    eng.sync()
    t = np.zeros((rounds, 10))
    for r in range(rounds):
        for k in range(10):
            t0 = time.perf_counter(); eng.tick(1); eng.sync(); t[r, k] = (time.perf_counter() - t0) * 1e6
    m = np.median(t, axis=0)
    print('%-34s us per single-tick launch by tick %% 10: %s   (sum of 10: %.0f)' % (name, ' '.join('%6.1f' % v for v in m), m.sum()), flush=True)
    eng.close()
base = dict(arena_size=250, num_pellets=500, num_viruses=10, mode=0, dt=1.0 / 60)
run('Tick/0', num_agents=0, example_bots=0, **base)
run('C1 (agent + 4 bot kinds)', num_agents=1, num_bots=4, **base)
run('agent + 4 ExampleBots', num_agents=1, example_bots=4, **base)
run('5 agents', num_agents=5, **base)
run('Tick/10', num_agents=0, example_bots=10, **base)
run('Tick/30', num_agents=0, example_bots=30, **base)
