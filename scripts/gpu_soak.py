"""One-off soak on the GPU: random configurations x random launch pins (lanes per arena, layout, single / two-kernel step), batched lock-step
against the oracle.  Prints the first mismatch or a summary.  (A raised capacity flag -- e.g. AGARCL_F_EVENTS_OVERFLOW in an 80 x 80 arena
with 1300 pellets, where a cell can eat more than 256 pellets in one tick -- is reported like a mismatch: that arena has left the
reference's unbounded containers by design.)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from agarcl_amd import _capi
from oracle import orabind
from lockstep import run_batched_lockstep
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = flagged = ref_undefined = 0
by_flag = {}     # flag word -> trials that ended with it: a regression in the capacity corner shows as a changed histogram, not only as a count
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    na = int(rng.choice([1, 1, 1, 1, 2, 3]))
    mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 6, 7, 8, 9, 10]))
    nb = int(rng.randint(0, 4)) if mode == 0 and rng.rand() < 0.4 else 0
    if mode > 6: na = 1
    cfg = dict(num_agents=na, arena_size=int(rng.choice([80, 150, 250, 400, 1000, 1100])), num_pellets=int(rng.choice([50, 64, 200, 500, 1000, 1300])),
               num_viruses=int(rng.choice([0, 0, 3, 10, 25])), num_bots=nb, mode=mode, reward_type=int(rng.randint(0, 2)), c_death=int(rng.choice([0, -20])))
    tps = 4
    if os.environ.get("SOAK_WIDE"):   # (as scripts/cpu_soak.py)
        cfg["arena_size"] = int(rng.choice([60, 120, 200, 300, 600, 1000, 1400])); cfg["ticks_per_step"] = tps = int(rng.choice([1, 2, 4, 8]))
        if rng.rand() < 0.2: cfg["pellet_regen"] = False
    if os.environ.get("SOAK_MANY"):
        cfg["mode"] = int(rng.choice([0, 0, 0, 4, 6])); cfg["num_agents"] = int(rng.randint(1, 7)); cfg["num_bots"] = int(rng.randint(0, 7)) if cfg["mode"] == 0 else 0
        cfg["arena_size"] = int(rng.choice([150, 250, 400]))
        if rng.rand() < 0.4: cfg["example_bots"] = int(rng.randint(1, 20))     # the reference's ExampleBots beside them: up to 32 players per arena
    pins = dict(AGARCL_TILE_LG=str(rng.choice([0, 6])), AGARCL_FUSED=str(rng.choice([0, 1])), AGARCL_FUSED_QG=str(rng.choice([1, 2, 4, 8, 16, 16])), AGARCL_QUIET_QG=str(rng.choice([1, 2, 4, 8, 16])),
                AGARCL_KSTEP_GRID=str(rng.choice([4096, 4096, 7, 32])), AGARCL_NO_ORDER=str(rng.choice([0, 0, 1])))     # (a small grid: several arenas per workgroup -- work counter, cost order)
    if os.environ.get('SOAK_NOPINS'): pins = {k: '' for k in pins}
    if os.environ.get('SOAK_TILE') is not None: pins['AGARCL_TILE_LG'] = os.environ['SOAK_TILE']
    for k_ in list(pins):
        if os.environ.get('SOAK_PIN_' + k_) is not None: pins[k_] = os.environ['SOAK_PIN_' + k_]
    os.environ.update(pins)
    only = int(os.environ.get('SOAK_ONLY', -1))
    A = int(rng.choice([3, 70, 130]))
    try:
        eng = _capi.BatchedEngine(A if only < 0 or only == trial else 1, **cfg)
    except _capi.AgarclError as e:   # e.g. squared pellets of a big arena exceed the pellet capacity: a loud rejection, not a case
        print('skipped', cfg, e); continue
    sd, ps, st = rng.randint(1, 1 << 30, size=A), int(rng.randint(1, 1000)), int(rng.choice([1, 4, 8]))
    if only >= 0 and only != trial: eng.close(); continue
    if os.environ.get('SOAK_VERBOSE'): print('trial', trial, cfg, pins, A, flush=True)
    oras = [orabind.OraEnv(**cfg) for _ in range(A)]
    ok, msg = run_batched_lockstep(eng, oras, 120, seeds=sd, policy_seed=ps, sticky=st, every=int(os.environ.get('SOAK_EVERY', 30)), ticks_per_step=tps)
    fl = eng.flags(); eng.close()
    if fl.any():   # an arena left the reference's unbounded containers / tables: flagged by design, not a parity failure
        w_ = int(np.bitwise_or.reduce(fl))
        # Whose limit was it?  In every flagged arena look at the REFERENCE side (the oracle kept running): an anti-team decay rate 0.002 * 1.1^(n-1) >= 1
        # (the player's stored 1.1^(n-1) >= 500: >= 67 viruses eaten within 3600 ticks -- a tiny arena full of viruses) makes the decay factor negative
        # and the reference's own unsigned mass wrap to ~2^32 (Engine.hpp:550-584, Entities.hpp:199-203); from there it grows cells without bound:
        # nothing an engine can follow.  (The wrapped cells themselves may be gone again by the checkpoint at which the flag is seen.)
        from oracle import blob
        und = True
        for a in np.nonzero(fl)[0]:
            pl = blob.parse(oras[int(a)].dump())["players"]
            und = und and any((p_["anti_team"] >= 500.0 or (p_["n_cells"] and int(p_["cell_mass"].max()) >= (1 << 31))) for p_ in pl)
        if und:
            ref_undefined += 1; print("reference left its domain (wrapped cell mass) trial", trial, cfg, "flags 0x%x" % w_); continue
        by_flag[w_] = by_flag.get(w_, 0) + 1
        flagged += 1; print("flagged (capacity) trial", trial, cfg, "flags 0x%x" % w_); continue
    if not ok:
        bad += 1; print("MISMATCH trial", trial, cfg, pins, A, msg); break
print("soak done:", trial + 1, "trials,", bad, "bad,", ref_undefined, "reference-undefined,", flagged, "flagged", "by flag word: " + ", ".join("0x%x: %d" % kv for kv in sorted(by_flag.items())) if by_flag else "")
