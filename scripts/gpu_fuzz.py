"""Long differential fuzz of the HIP engine against the C oracle (run through gpurun): gpu_fuzz.py <trials> <seed>"""
import sys, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from agarcl_amd import _capi
from oracle import orabind
from lockstep import run_batched_lockstep, run_quiet_rollout
orabind.build()
trials, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
bad = 0
for trial in range(trials):
    na = int(rng.choice([1, 1, 1, 2, 3, 5]))
    mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 6, 7, 8, 9, 10]))
    nb = int(rng.randint(0, 6)) if mode == 0 else 0
    if mode > 6: na = 1
    cfg = dict(num_agents=na, arena_size=int(rng.choice([60, 80, 150, 250, 400, 1000, 1020, 1021, 1100, 2000])), num_pellets=int(rng.choice([1, 50, 64, 65, 200, 256, 500, 1000, 1024, 1300, 2048])),
               num_viruses=int(rng.choice([0, 1, 3, 10, 25, 60])), num_bots=nb, mode=mode, reward_type=int(rng.randint(0, 2)), c_death=int(rng.choice([0, -20])))
    A = int(rng.choice([1, 3, 4, 5, 9]))
    try:
        eng = _capi.BatchedEngine(A, **cfg)
    except Exception as ex:
        print('trial', trial, cfg, 'create failed:', ex); continue
    oras = [orabind.OraEnv(**cfg) for _ in range(A)]
    seeds = rng.randint(1, 1 << 30, size=A)
    if rng.rand() < 0.35 and na == 1 and nb == 0 and mode <= 6:
        ok, msg = run_quiet_rollout(eng, oras, 400, seeds, rng_seed=int(rng.randint(1, 1000)), check_every=50)
    else:
        ok, msg = run_batched_lockstep(eng, oras, 200, seeds=seeds, policy_seed=int(rng.randint(1, 1000)), sticky=int(rng.choice([1, 4, 8])), every=10)
    fl = eng.flags(); eng.close()
    if not ok:
        bad += 1; print('MISMATCH trial', trial, cfg, 'A', A, msg, 'flags', fl.tolist(), flush=True)
print('fuzz done: %d trials, %d mismatches' % (trials, bad))
