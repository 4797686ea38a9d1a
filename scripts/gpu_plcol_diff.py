"""players_collision applied from the lanes' scan records (product) against the lane-0 replay alone (-DAG_PLCOL_REPLAY_ONLY ->
build_variants/lib_PLOLD.so): the same arenas stepped by both libraries, masses / counts / rewards compared every step and whole arenas at the end.
python scripts/gpu_plcol_diff.py"""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from agarcl_amd import _capi
old = _capi.bind(C.CDLL('build_variants/lib_PLOLD.so'))
def run(name, A, steps, **cfg):
    na = cfg.get('num_agents', 1)
    e1 = _capi.BatchedEngine(A, **cfg); e2 = _capi.BatchedEngine(A, lib=old, **cfg)
    for e in (e1, e2): e.seed(None, 77); e.reset(reset_ids=True)
    rng = np.random.RandomState(1)
    mv = [rng.uniform(-1, 1, size=(A, max(na, 1), 2)).astype(np.float32) for _ in range(8)]
    ac = [rng.randint(0, 3, size=(A, max(na, 1))).astype(np.int32) for _ in range(8)]
    bad = 0; eats = 0; prev = None
    for k in range(steps):
        for e in (e1, e2):
            if na: e.set_actions(mv[k % 8], ac[k % 8]); e.step(4)
            else: e.tick(4)
        c1, c2 = e1.counts(), e2.counts()
        if prev is not None: eats += int((c1[:, 3] < prev).sum())
        prev = c1[:, 3].copy()
        if not (np.array_equal(c1, c2) and (na == 0 or (np.array_equal(e1.masses(), e2.masses()) and np.array_equal(e1.rewards(), e2.rewards())))):
            bad += 1
            if bad <= 3: print('   step %d differs in arenas %s' % (k, np.nonzero((c1 != c2).any(axis=1))[0][:8]))
    whole = sum(1 for a in range(0, A, max(1, A // 64)) if bytes(e1.dump(a)) != bytes(e2.dump(a)))
    print('%-44s A=%d steps=%d: steps that differ %d, sampled whole arenas that differ %d, arena-steps in which the cell count fell %d, flags %s / %s' % (
        name, A, steps, bad, whole, eats, sorted(set(int(f) for f in e1.flags() if f)), sorted(set(int(f) for f in e2.flags() if f))), flush=True)
    e1.close(); e2.close()
run('C1 (agent + 4 bot kinds)', 2048, 1200, num_agents=1, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60)
run('5 agents mode 6, 150x150', 1024, 500, num_agents=5, arena_size=150, num_pellets=300, num_viruses=0, mode=6)
run('3 agents + 6 bots, 120x120', 1024, 800, num_agents=3, arena_size=120, num_pellets=300, num_viruses=4, num_bots=6, mode=0)
run('2 agents + 12 bots + 8 ExampleBots, 200x200', 512, 800, num_agents=2, arena_size=200, num_pellets=400, num_viruses=5, num_bots=12, example_bots=8, mode=0)
run('Tick/20 with 4 agents mode 6', 512, 400, num_agents=4, arena_size=250, num_pellets=500, num_viruses=10, example_bots=20, mode=6, dt=1.0 / 60)
