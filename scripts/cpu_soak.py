"""CPU soak: the kernel source compiled as a host emulation (tests/_build/libagarcl_emu.so) against the oracle on random configurations,
every trial in its own process under a timeout (a hang is a finding too).   python scripts/cpu_soak.py <seed> <trials>"""
import multiprocessing as mp, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def trial(args):
    seed, k = args
    import ctypes, numpy as np
    from agarcl_amd import _capi
    from oracle import orabind
    from lockstep import run_batched_lockstep
    rng = np.random.RandomState(seed * 1000 + k)
    na = int(rng.choice([1, 1, 1, 1, 2, 3]))
    mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 6, 7, 8, 9, 10]))
    nb = int(rng.randint(0, 4)) if mode == 0 and rng.rand() < 0.4 else 0
    if mode > 6: na = 1
    cfg = dict(num_agents=na, arena_size=int(rng.choice([80, 150, 250, 400, 1000, 1100])), num_pellets=int(rng.choice([50, 64, 200, 500, 1000, 1300])),
               num_viruses=int(rng.choice([0, 0, 3, 10, 25])), num_bots=nb, mode=mode, reward_type=int(rng.randint(0, 2)), c_death=int(rng.choice([0, -20])))
    if os.environ.get("SOAK_WIDE"):   # a wider configuration space: other arena sizes, frame skips, tick lengths, no regeneration
        cfg["arena_size"] = int(rng.choice([60, 120, 200, 300, 600, 1000, 1400])); cfg["ticks_per_step"] = int(rng.choice([1, 2, 4, 8]))
        rng.rand()   # (dt = 1/60 belongs to the engine-level tick path, tests/lockstep.py run_engine_level_lockstep: BaseEnvironment::step has its own tick length)
        if rng.rand() < 0.2: cfg["pellet_regen"] = False
    if os.environ.get("SOAK_MANY"):   # crowded arenas: up to 6 agents and 6 bots (mode 0)
        cfg["mode"] = int(rng.choice([0, 0, 0, 4, 6])); cfg["num_agents"] = int(rng.randint(1, 7)); cfg["num_bots"] = int(rng.randint(0, 7)) if cfg["mode"] == 0 else 0
        cfg["arena_size"] = int(rng.choice([150, 250, 400]))
        if rng.rand() < 0.4: cfg["example_bots"] = int(rng.randint(1, 20))     # the reference's ExampleBots beside them: up to 32 players per arena
    engine_level = bool(os.environ.get("SOAK_ENGINE")) and rng.rand() < 0.7   # bench/main.cpp's path: Engine::tick at dt = 1/60, respawns
    if engine_level: cfg["dt"] = 1.0 / 60
    os.environ["AGARCL_TILE_LG"] = str(rng.choice([0, 6]))
    lib = _capi.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "_build", "libagarcl_emu.so")))
    A = int(rng.choice([2, 3, 5]))
    try:
        eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    except _capi.AgarclError as e:
        return ("skipped", cfg, str(e))
    oras = [orabind.OraEnv(**cfg) for _ in range(A)]
    if engine_level:
        from lockstep import run_engine_level_lockstep
        ok, msg = run_engine_level_lockstep(eng, oras, int(rng.choice([400, 1200, 2500])), seeds=rng.randint(1, 1 << 30, size=A), policy_seed=int(rng.randint(1, 1000)), every=20)
    else:
      ok, msg = run_batched_lockstep(eng, oras, int(rng.choice([120, 300, 600])), seeds=rng.randint(1, 1 << 30, size=A), policy_seed=int(rng.randint(1, 1000)),
                                     sticky=int(rng.choice([1, 4, 8])), every=10, ticks_per_step=cfg.get("ticks_per_step", 4))
    fl = eng.flags(); eng.close()
    if fl.any():
        # whose limit?  (as scripts/gpu_soak.py: a cell mass above 2^31 on the REFERENCE side = its unsigned arithmetic went below zero)
        from oracle import blob
        und = all(any((p_["anti_team"] >= 500.0 or (p_["n_cells"] and int(p_["cell_mass"].max()) >= (1 << 31))) for p_ in blob.parse(oras[int(a)].dump())["players"]) for a in np.nonzero(fl)[0])
        return ("reference-undefined" if und else "flagged", cfg, "0x%x" % int(np.bitwise_or.reduce(fl)))
    return ("ok", cfg, "") if ok else ("MISMATCH", cfg, msg)


if __name__ == "__main__":
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    counts = {}
    with mp.get_context("spawn").Pool(int(os.environ.get("SOAK_PROCS", 6)), maxtasksperchild=1) as pool:
        results = [pool.apply_async(trial, ((seed, k),)) for k in range(n)]
        for k, r in enumerate(results):
            try:
                kind, cfg, msg = r.get(timeout=600)
            except mp.TimeoutError:
                kind, cfg, msg = "HANG", "trial %d" % k, ""
            counts[kind] = counts.get(kind, 0) + 1
            if kind in ("MISMATCH", "HANG", "flagged"): print(kind, "seed", seed, "trial", k, cfg, msg, flush=True)   # (flagged trials with their flag word: the capacity corner)
    print("cpu soak seed", seed, counts, flush=True)
