#!/usr/bin/env python
"""Headline benchmark: env-steps/s (arenas x engine ticks per second) of the batched HIP engine.

Workload = BASELINE.json configs[1] / SURVEY.md 8(d) "C2": 4096 arenas per GPU, each 1000x1000,
1000 pellets, 0 viruses, 1 agent, no bots, pellet regen, mode 0, dt = 1/30, 4 ticks per env step,
action "none", (dx,dy) ~ U(-1,1)^2 pre-generated in HBM, arena seeds = 10000 + global arena index.
A bench "step" is one agarcl_step = 4 engine ticks of every arena.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python bench.py --gpus 8 ...        (WORLD_SIZE unset: starts the 8 rank processes itself through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (DESIGN.md section 5 explains the roofline / cpu_baseline fields).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARENAS_PER_GPU = 4096
LARGE_ARENAS = 65536        # the "roofline_large" block: north star ">= 50k parallel arenas"
XLARGE_ARENAS = 262144      # "roofline_xlarge": the same kernels where they are bandwidth-bound (two-kernel step, one lane per arena)
SWEEP_ARENAS = (16384, 131072)   # "roofline_sweep": two more sizes around roofline_large
CFG = dict(num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000, num_viruses=0,
           num_bots=0, reward_type=1, c_death=0, mode_number=0)
# The headline line is C2.  The other SURVEY 8(d) workloads are selectable for DESIGN.md's measurement table only.
WORKLOADS = {
    "C2": dict(desc="C2: %d arenas/GPU x 1 agent, 1000x1000, 1000 pellets, 0 viruses, mode 0, 4 ticks/step, random (dx,dy), action none"),
    "C3m0": dict(num_viruses=25, rand_act=True,
                 desc="C3/mode 0: %d arenas/GPU x 1 agent, 1000x1000, 1000 pellets, 25 viruses, mode 0, 4 ticks/step, random (dx,dy), action ~ U{0,1,2}"),
    "C3m6": dict(num_viruses=25, mode_number=6, rand_act=True,
                 desc="C3/mode 6: %d arenas/GPU x 1 agent (mass 1000), 1000x1000, 1000 pellets, 25 viruses, mode 6, 4 ticks/step, random (dx,dy), action ~ U{0,1,2}"),
    "C5": dict(num_viruses=25, mode_number=6, rand_act=True, grid_obs=True,
               desc="C5: C3/mode 6 + int32 grid observation [%d][8][128][128] refreshed once per step (persistent HBM tensor)"),
    "C5s": dict(num_viruses=25, mode_number=6, rand_act=True, screen_obs=True,
                desc="C5 (screen): C3/mode 6 + uint8 screen observation [%d][84][84][3] written once per step"),
    # between the quiet headline and the mass-1000 extreme: what a learning agent's mid-game costs
    "mid": dict(num_viruses=25, rand_act=True, start_mass=150,
                desc="mid-game: %d arenas/GPU x 1 agent grown to mass 150 (ejects / splits fire, ejected food lies around), 1000x1000, 1000 pellets, 25 viruses, mode 0, 4 ticks/step, action ~ U{0,1,2}"),
    # BASELINE configs[0] batched: the reference's own bench/main.cpp population (agent + the four bot kinds on the default
    # 250x250 arena, Engine::tick at dt = 1/60 s), 4 ticks per launch
    "C1": dict(arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, dt=1.0 / 60, rand_act=True,
               desc="C1 batched: %d arenas/GPU x (1 agent + 4 bots), 250x250, 500 pellets, 10 viruses, mode 0, dt 1/60 s, 4 ticks/step"),
    # the 32-slot several-player instantiation (k_step<32, *, *, MP = true>: more than 1024 pellets with several players; VERDICT r5 weak #1e)
    "P5big": dict(arena_size=400, num_pellets=1500, num_viruses=10, num_bots=4, rand_act=True,
                  desc="P5big: %d arenas/GPU x (1 agent + 4 bots), 400x400, 1500 pellets, 10 viruses, mode 0, 4 ticks/step"),
    # ... and the single-player one (more than 1024 pellets under a mass-1000 agent)
    "C3m6big": dict(num_pellets=1500, num_viruses=25, mode_number=6, rand_act=True,
                    desc="C3m6big: %d arenas/GPU x 1 agent (mass 1000), 1000x1000, 1500 pellets, 25 viruses, mode 6, 4 ticks/step"),
    "C1r": dict(arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, dt=1.0 / 60, rand_act=True, ram_obs=True,
                desc="C1 batched + ram observation: %d arenas/GPU x (1 agent + 4 bots), 250x250, 500 pellets, 10 viruses, mode 0, dt 1/60 s, 4 ticks/step, f32 [A][1][152] written once per step"),
}
TRAFFIC_FILE = "r06_pmc_traffic.json"   # PMC FETCH_SIZE / WRITE_SIZE (+ one SQ pass) per step of the bench workloads, recorded by scripts/profile_round.sh
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)
GUIDE_COPY_GBS = 6300.0   # the achievable device copy rate the guide quotes (MI355X_MICROARCH.md), beside the copy timed in this process
PIPE_K = 4                # sub-batches of the "<workload>/pipe4" entries (agarcl_pipe_*: independent arena ranges on streams of their own)
TASKS_FIXTURE = os.path.join(ROOT, "tests", "golden", "paper_tasks.json")   # values of the reference's bench/tasks_configs/mode_{1..10}.json


def paper_tasks():
    """{m: config values} of the reference's ten RL tasks (tests/golden/make_tasks_fixture.py wrote them from /root/reference/bench/tasks_configs)"""
    try:
        return {int(k): v for k, v in json.load(open(TASKS_FIXTURE))["tasks"].items()}
    except Exception:
        return {}


def task_workload(m):
    """bench workload of task m: the engine arguments + the 128 x 128 agent-view frame written every step (obs_type "screen": the screen env's
    respawn hook is on, ScreenEnvironment.hpp:233-243)"""
    t = paper_tasks()[m]
    cfg = dict(num_agents=1, ticks_per_step=t["ticks_per_step"], arena_size=t["arena_size"], pellet_regen=bool(t["pellet_regen"]), num_pellets=t["num_pellets"],
               num_viruses=t["num_viruses"], num_bots=t["num_bots"], reward_type=t["reward_type"], c_death=t["c_death"], mode_number=t["mode"], screen_respawn=True)
    desc = ("task %d (bench/tasks_configs/mode_%d.json): %%d arenas/GPU, %dx%d arena, %d pellets, %d viruses, %d bot(s), mode %d, %d ticks/step, action ~ U{0,1,2}, "
            "uint8 agent-view frame [%%d][%d][%d][4] written every step" % (m, m, t["arena_size"], t["arena_size"], t["num_pellets"], t["num_viruses"], t["num_bots"],
                                                                               t["mode"], t["ticks_per_step"], t["screen_len"], t["screen_len"]))
    return cfg, (t["screen_len"], t["screen_len"], bool(t["agent_view"])), desc


def source_sha():
    """Identifies the kernel source a profile was recorded on (profiles/*.json carry it; a stale one is not quoted)."""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "agarcl_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def requested_bytes(work, counts, P, n_agents, ticks, pellet_cap):
    """Bytes the step kernels REQUEST from memory, from the kernels' own work counters (agarcl_debug_work) -- what this
    implementation's algorithm needs, as opposed to the streaming model of SURVEY 8(d) (every entity once per tick).
    Never more than the HBM traffic the PMC counters see (requests are rounded up to 64/128-byte sectors), so the
    fraction built on it cannot exceed 1.  Constants = the loads / stores written in agar_quiet.inl (front part) and
    agar_core.inl arena_load / arena_store (general engine); the shared mass tables stay in L2 and are not counted."""
    front_steps, general_steps, pellet_moves = (float(x) for x in work[:3])
    n_pel, n_vir, n_food, n_cells = counts
    # front part, per arena-step: loads cell 0 (48) + 18 player words (72) + 9 arena words (36) + action (12);
    # stores cell (40) + player (64) + arena (40) + counts (16) + results f64/i32/u8/packed (21) + hand-over (8)
    front = 168.0 + 189.0
    # general engine, per arena-step: arena words r+w (256), player words r+w (192 P), all 32 cell slots read (1536 P) with one or two players,
    # only the live cells (48 N_c) with more (r05: arena_load) + live cells written (48 N_c), results (29 A) + counts (16) + its cycle count (4);
    # viruses and foods are staged in LDS ONCE per launch
    # (round 4): x / y / mass of the whole virus table (12 (N_v + 64): the capacity, no count is known when the loads are issued), the first
    # 64 foods (x, y, vx, vy: 1024) read, the live foods written back (16 N_f)
    general = 256.0 + 192.0 * P + (1536.0 * P if P <= 2 else 48.0 * n_cells) + 48.0 * n_cells + 29.0 * n_agents + 20.0 + 12.0 * (round(n_vir) + 64.0) + 1024.0 + 16.0 * n_food
    return front * front_steps + general * general_steps + pellet_moves * pellet_cap * 8.0


def cpu_baseline(seconds_budget=10.0, c3_budget=5.0):
    """The same per-arena workload on the host cores, timed on a bounded sample.
    kind "reference": oracle/_ref/libagar_ref.so = the unmodified reference engine (prebuilt from
    /root/reference); otherwise kind "port": the plain-C restatement."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota (containers report the host's CPU count otherwise)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per))))
    except Exception:
        pass
    try:
        from oracle import refbind
        use_ref = refbind.available()
    except Exception:
        use_ref = False
    if use_ref:
        from oracle import refbind as B
        Env, kind = B.RefEnv, "reference"
    else:
        from oracle import orabind as B
        if not B.available():
            B.build()
        Env, kind = B.OraEnv, "port"
    mk = lambda: Env(num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                     num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode=0)
    # calibrate on one core
    e = mk(); e.seed(10000); e.reset(True)
    t0 = time.perf_counter(); e.run_random(4000, policy_seed=1, allow_actions=False); dt1 = time.perf_counter() - t0
    rate1 = 4000 / dt1
    # BASELINE configs[0] as bench/main.cpp:14-38 runs it: default engine (250x250, 500 pellets, 10 viruses, mode 0), agent +
    # the four bot kinds, dt = 1/60 s, random policy, 10 000 ticks, one core
    c1 = Env(num_agents=1, ticks_per_step=4, arena_size=250, pellet_regen=True, num_pellets=500, num_viruses=10, num_bots=4,
             reward_type=1, c_death=0, mode=0, dt=1.0 / 60)
    c1.seed(42); c1.reset(True)
    t0 = time.perf_counter(); n1 = c1.run_random(10000, policy_seed=7, allow_actions=True); c1_rate = n1 / (time.perf_counter() - t0)
    c1.close()
    chunk = 2000
    done = [0] * cores
    mk_lock = threading.Lock()
    deadline = time.perf_counter() + seconds_budget

    def worker(w):  # one engine per thread at a time (the reference's BotEvaluator pattern), time-bounded
        j = 0
        while time.perf_counter() < deadline:
            with mk_lock:
                env = mk()
            env.seed(10000 + w * 64 + j); env.reset(True)
            for _ in range(10):
                done[w] += env.run_random(chunk, policy_seed=w * 64 + j + 1, allow_actions=False)
                if time.perf_counter() >= deadline:
                    break
            env.close(); j += 1
    th = [threading.Thread(target=worker, args=(w,)) for w in range(cores)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    total = sum(done)
    # the full rule set beside it (BASELINE configs[2] = C3, mode 6: agent mass 1000, 25 viruses, action ~ U{0,1,2}): same engine, all cores,
    # a bounded 6 s
    mk6 = lambda: Env(num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                      num_viruses=25, num_bots=0, reward_type=1, c_death=0, mode=6)
    done6 = [0] * cores
    deadline6 = time.perf_counter() + c3_budget

    def worker6(w):
        j = 0
        while time.perf_counter() < deadline6:
            with mk_lock:
                env = mk6()
            env.seed(10000 + w * 64 + j); env.reset(True)
            for _ in range(20):
                done6[w] += env.run_random(400, policy_seed=w * 64 + j + 1, allow_actions=True)
                if time.perf_counter() >= deadline6:
                    break
            env.close(); j += 1
    th = [threading.Thread(target=worker6, args=(w,)) for w in range(cores)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el6 = time.perf_counter() - t0
    return {"value": total / el, "unit": "env-steps/s", "cores": cores, "kind": kind,
            "c3m6_value": sum(done6) / el6, "c3m6_cores": cores,
            "c3m6_sample": "%.0f s of C3 / mode 6 (1000x1000, 1000 pellets, 25 viruses, agent mass 1000, random dx/dy, action ~ U{0,1,2}), one engine per host "
                           "thread, %d threads, %d arena-ticks in total" % (c3_budget, cores, sum(done6)),
            "sample": "%.0f s of C2 on the host: one engine per thread, %d threads, %d arena-ticks; one thread alone: %.0f ticks/s"
                      % (seconds_budget, cores, total, rate1),
            "c1_ticks_per_s_1core": c1_rate,
            "c1_sample": "BASELINE configs[0] as bench/main.cpp runs it: 250x250, 500 pellets, 10 viruses, agent + 4 bot kinds, "
                         "dt 1/60 s, random policy, 10000 ticks on one core"}


TICK_BOTS = (0, 5, 10, 20, 30)   # bench/main.cpp:38  BENCHMARK(Tick)->Arg(0)->Arg(5)->Arg(10)->Arg(20)->Arg(30)


def tick_population(n_bots, dt=1.0 / 60):
    """bench/main.cpp:14-24: a default engine (250 x 250, 500 pellets, 10 viruses) with N ExampleBots and no Player, ticked at dt = 1/60 s"""
    return dict(num_agents=0, ticks_per_step=4, arena_size=250, pellet_regen=True, num_pellets=500, num_viruses=10, num_bots=0, reward_type=1,
                c_death=0, mode=0, dt=dt, example_bots=n_bots)


def run_tick_workload(n_bots, A, K, Wm, device=0):
    """The literal bench/main.cpp Tick/N population batched: A arenas, K timed launches of 4 engine ticks each (agarcl_tick: no env around it)."""
    import numpy as np
    from agarcl_amd import _capi
    cfg = tick_population(n_bots)
    eng = _capi.BatchedEngine(A, device=device, **cfg)
    eng.seed(None, 10000); eng.reset(reset_ids=True)
    for _ in range(Wm):
        eng.tick(4)
    eng.work(reset=True)
    eng.timer_mark(0); eng.timer_mark(1); eng.sync()
    t0 = time.perf_counter()
    eng.timer_mark(0)
    for _ in range(K):
        eng.tick(4)
    eng.timer_mark(1)
    eng.sync()
    elapsed = time.perf_counter() - t0
    res = dict(elapsed=elapsed, kernel_ms=eng.timer_elapsed_ms() / K, work=eng.work(), flags=eng.flags(), counts=eng.counts().astype(np.float64).mean(axis=0),
               players=eng.players, fused=0, pellet_cap=(cfg["num_pellets"] + 63) // 64 * 64, ranks=[])
    eng.close()
    return res, cfg


def tick_reference_rate(n_bots, seconds=0.6):
    """the same population on the reference engine, one host core: engine ticks per second (oracle/_ref, kind "reference"; else the C port)"""
    try:
        from oracle import refbind as B
        if not B.available():
            raise ImportError
        env, kind = B.RefEnv(**tick_population(n_bots)), "reference"
    except Exception:
        from oracle import orabind as B
        if not B.available():
            B.build()
        env, kind = B.OraEnv(**tick_population(n_bots)), "port"
    env.seed(42); env.reset(True)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            env.tick()
        n += 200
    rate = n / (time.perf_counter() - t0)
    env.close()
    return rate, kind


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (before this process has made any GPU
    call) and return their exit code; never silently run fewer ranks than asked for."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def run_workload(torch, np, env_cls, agdist, dev, dev_index, rank, world, A, K, Wm, cfg, rand_act, with_obs, with_screen,
                 gather_mode, gather_obs, gather_block=32, start_mass=0, with_ram=False, sub_batches=1, screen=(84, 84, False)):
    """Builds the env, runs Wm untimed + K timed steps, returns the measurements of this rank.
    sub_batches > 1: the rank's arenas as that many independent sub-batches on HIP streams of their own (agarcl_pipe_*): every step enqueues
    all of them, none waits for another's slowest arena, one's observation kernel runs under another's step.  Same arenas, same results."""
    import torch.distributed as dist
    lo, hi = rank * A, (rank + 1) * A  # weak scaling: every GPU owns `A` arenas
    if sub_batches > 1:
        from agarcl_amd.vec_env import PipelinedVecEnvironment
        penv = PipelinedVecEnvironment(A, sub_batches, device=dev_index, strict_flags=False, **cfg)
        penv.seed(agdist.arena_seeds(10000, lo, hi)); penv.reset(reset_ids=True)
        parts, ranges, env = penv.parts, penv.ranges, penv.parts[0]
    else:
        penv = None
        env = env_cls(A, device=dev_index, strict_flags=False, **cfg)
        env.seed(agdist.arena_seeds(10000, lo, hi))
        env.reset(reset_ids=True)
        parts, ranges = [env], [(0, A)]
    if start_mass:   # grown agents, through the product's own snapshot path (the reference's JSON wire format, agarcl_amd/snapshot.py)
        from agarcl_amd import snapshot
        sn = snapshot.save_arena(env.engine, 0, cfg)   # one start state for every arena (seeds and actions differ): enough for a timing
        for pl in sn["players"]:
            for cell in pl["cells"]:
                cell["mass"] = int(start_mass)
        for p_, (l_, n_) in zip(parts, ranges):
            for a in range(n_):
                sn["seed"] = 10000 + lo + l_ + a
                snapshot.load_arena(p_.engine, a, sn, reset_ids=True)
    # synthetic random policy, resident in HBM before the timed region: counter-based per (arena, step)
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    na = cfg["num_agents"]
    dxdy = (torch.rand((K + Wm, A, na, 2), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    act = torch.zeros((K + Wm, A, na), dtype=torch.int32, device=dev)
    if rand_act:
        act = torch.randint(0, 3, (K + Wm, A, na), generator=g, device=dev, dtype=torch.int32)
    obs = torch.empty((A, 8, 128, 128), dtype=torch.int32, device=dev) if with_obs else None
    want_screen = with_screen or (world > 1 and gather_obs == "screen")
    sw, sh, sav = screen
    scr = torch.empty((A, sh, sw, 4 if sav else 3), dtype=torch.uint8, device=dev) if want_screen else None
    ram = torch.empty((A, na, 152), dtype=torch.float32, device=dev) if with_ram else None
    torch.cuda.synchronize()                    # (sub-batches run on streams of their own: the policy tensors are complete before they start)
    # Multi-GPU result path (the only exchange there is: arenas never interact).
    #   block: (reward, done) of 32 (--gather-block) consecutive steps -- one contiguous block of the engine's 64-slot result
    #          ring, zero copy -- per asynchronous RCCL gather, double-buffered by ring half: a rollout chunk, as an n-step learner
    #          consumes them; the collective's host-side launch cost is paid once per block, not once per 9 us step
    #   step : one gather per step straight from the slot the step wrote (what a learner on rank 0 needs; ~20 us of latency each)
    #   --gather-obs screen: additionally every step's uint8 frames [A][84][84][3] go to rank 0 (21 KB per arena)
    BLK = int(gather_block)
    SLOTS = 64                                  # agarcl_batch.h AGARCL_PACKED_SLOTS
    assert SLOTS % (2 * BLK) == 0
    eng = env.engine
    engs = [p_.engine for p_ in parts]
    gathers = obs_gather = None
    if world > 1:
        # one gatherer per sub-batch (its (reward, done) ring is its own engine's memory: zero copy)
        gathers = [agdist.ResultGatherer(BLK * n_ * na if gather_mode == "block" else n_ * na, dev, depth=2) for (_, n_) in ranges]
        if gather_obs == "screen":
            obs_gather = agdist.TensorGatherer((A, sh, sw, 4 if sav else 3), torch.uint8, dev)

    # raw pointers of the pre-generated policy output, one pair per step and sub-batch (no tensor indexing inside the timed loop)
    dx_ptr = [[dxdy[k].data_ptr() + l_ * na * 8 for (l_, _) in ranges] for k in range(K + Wm)]
    ac_ptr = [[act[k].data_ptr() + l_ * na * 4 for (l_, _) in ranges] for k in range(K + Wm)]
    obs_ptr = [obs.data_ptr() + l_ * 8 * 128 * 128 * 4 for (l_, _) in ranges] if obs is not None else None
    scr_ptr = [scr.data_ptr() + l_ * sh * sw * (4 if sav else 3) for (l_, _) in ranges] if scr is not None else None
    ram_ptr = [ram.data_ptr() + l_ * na * 152 * 4 for (l_, _) in ranges] if ram is not None else None
    tps = cfg["ticks_per_step"]
    slot_of = lambda k: (first_slot + k) % SLOTS   # the ring slot step k writes (every engine's slot counter advances by one per step)
    first_slot = (eng.last_slot() + 1) % SLOTS
    piped = penv is not None

    def one_step(k):
        if gathers is not None and gather_mode == "block":
            nxt = slot_of(k)
            if nxt % BLK == 0:
                for g_ in gathers:
                    g_.wait_slot((nxt // BLK) & 1)   # the engine is about to write this block: the gather that last used its buffer has left
        for j, e_ in enumerate(engs):
            e_.step_actions(dx_ptr[k][j], ac_ptr[k][j], tps)   # take_actions + step: one host call
            if obs is not None:
                e_.grid_obs(128, True, True, True, True, out_ptr=obs_ptr[j], persistent=True)   # the same tensor every step
            if ram is not None:
                e_.ram_obs(16, 16, 8, 16, out_ptr=ram_ptr[j])
            if scr is not None:
                if obs_gather is not None and j == 0:
                    obs_gather.wait()           # the previous step's frames have left before they are overwritten
                    if piped:
                        for p_ in parts: p_.order_after_current()
                e_.screen_obs(sw, sh, out_ptr=scr_ptr[j], agent_view=sav)
        if obs_gather is not None:
            if piped:
                for p_ in parts: p_.order_current_after()
            obs_gather.gather(scr)
        if gathers is not None:
            s_ = slot_of(k)
            for j, g_ in enumerate(gathers):
                ring = parts[j].packed_ring
                if gather_mode == "step":
                    g_.wait_slot(s_ & 1)
                    if piped: parts[j].order_current_after()
                    g_.gather_packed(s_ & 1, ring[s_])
                elif s_ % BLK == BLK - 1:           # RCCL gather of one block of steps of (reward, done) straight from engine memory
                    h_ = s_ // BLK
                    if piped: parts[j].order_current_after()
                    g_.gather_packed(h_ & 1, ring[h_ * BLK:(h_ + 1) * BLK].reshape(-1, 2))

    def flush():                                # results of a partial last block still go to rank 0
        s_ = eng.last_slot()
        if gathers is not None and gather_mode == "block" and s_ % BLK != BLK - 1:
            h_ = s_ // BLK
            for j, g_ in enumerate(gathers):
                g_.wait_slot(h_ & 1)
                if piped: parts[j].order_current_after()
                g_.gather_packed(h_ & 1, parts[j].packed_ring[h_ * BLK:(h_ + 1) * BLK].reshape(-1, 2))

    def drain():
        for g_ in (gathers or []):
            g_.wait_all()
        if obs_gather is not None:
            obs_gather.wait()

    def sync_all():
        for e_ in engs:
            e_.sync()
        torch.cuda.synchronize()

    for k in range(Wm):
        one_step(k)
    flush(); drain()
    for e_ in engs:
        e_.work(reset=True)                     # (synchronising) the kernels' work counters restart with the timed region
    for g_ in (gathers or []) + [obs_gather]:
        if g_ is not None:
            g_.reset_stats()
    # HIP events on the launch stream(s) (the engine's own pair: created once, recorded without the system-scope fence of an ordinary
    # event record); marking them once here creates them outside the timed region
    for e_ in engs:
        e_.timer_mark(0); e_.timer_mark(1)
    sync_all()
    if world > 1:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    for e_ in engs:
        e_.timer_mark(0)
    for k in range(Wm, Wm + K):
        one_step(k)
    flush()
    for e_ in engs:
        e_.timer_mark(1)
    drain()
    sync_all()
    elapsed_own = time.perf_counter() - t0     # this rank's own time, before it waits for the others
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch stream (the engine adopts torch's current stream; a sub-batch has its own): avg per step, the slowest sub-batch
    kernel_ms = max(e_.timer_elapsed_ms() for e_ in engs) / K
    # per-rank record (so that a poor scaling curve can be read off the line): this rank's own wall time and step time by HIP events,
    # the host time it spent waiting for collectives, the payload bytes it contributed
    gl = gathers or []
    # bytes this rank's kernels requested per step, from ITS OWN work counters (the PMC traffic of profiles/ is a single-GPU recording and is
    # not quoted for N > 1 lines)
    own_work = sum(e_.work() for e_ in engs)
    own_counts = sum(e_.counts().astype(np.float64).sum(axis=0) for e_ in engs) / float(A)
    own_req = requested_bytes(own_work, own_counts, eng.players, cfg["num_agents"], cfg["ticks_per_step"], (cfg["num_pellets"] + 63) // 64 * 64) / K
    rank_info = {"rank": rank, "ms_per_step": elapsed_own / K * 1e3, "kernel_ms_per_step": kernel_ms, "requested_bytes_per_step": own_req,
                 "requested_GBs": own_req / (kernel_ms * 1e-3) / 1e9,
                 "gather_wait_ms_total": sum(g_.wait_s for g_ in gl) * 1e3 + (obs_gather.wait_s if obs_gather is not None else 0.0) * 1e3,
                 "result_collectives": sum(g_.calls for g_ in gl), "result_bytes_sent": sum(g_.bytes_sent for g_ in gl),
                 "obs_collectives": obs_gather.calls if obs_gather is not None else 0, "obs_bytes_sent": obs_gather.bytes_sent if obs_gather is not None else 0}
    ranks = [rank_info]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)
    counts = own_counts
    res = dict(elapsed=elapsed, kernel_ms=kernel_ms, ranks=ranks, world=world, work=own_work, flags=np.concatenate([e_.flags() for e_ in engs]),
               counts=counts, players=eng.players, fused=int(eng.L.agarcl_debug_fused(eng.h)),
               pellet_cap=(cfg["num_pellets"] + 63) // 64 * 64, sub_batches=len(engs), concurrent=(penv.concurrent if penv is not None else 1))
    (penv if penv is not None else env).close()
    return res


def obs_bytes(res, A, cfg, with_obs, with_screen, with_ram, screen=(84, 84, False)):
    """(requested, streaming-model) observation bytes per step.  Streaming model: the whole tensor is written once per step.
    Requested by this implementation: the grid tensor is persistent (agarcl_grid_obs on_device = 2), so per word scattered a step clears
    the old one and writes the new one plus their undo-list entries (16 B; the view covers at most (300 / arena)^2 of the arena's pellets
    and viruses), reads and writes the out-of-bounds channel's row / column signature (2 x 2 x 128 B) and stores the rows / columns of that
    channel whose signature byte changed (round 4; not counted: a few 512-byte rows per agent near a wall, none elsewhere)"""
    model_extra = (A * 8 * 128 * 128 * 4 if with_obs else 0) + (A * screen[0] * screen[1] * (4 if screen[2] else 3) if with_screen else 0) + (A * cfg["num_agents"] * 152 * 4 if with_ram else 0)
    extra = model_extra
    if with_obs:
        n_pel, n_vir, n_food, n_cells = res["counts"]
        vis = min(1.0, (300.0 / cfg["arena_size"]) ** 2)
        extra = A * (512.0 + 16.0 * (vis * 2 * (n_pel + n_vir) + 3 * n_cells))
    if with_ram:
        # k_ram_obs reads every live entity once to pick the K nearest (the pellets are not in registers there): those reads are what it requests
        # (round 5 counted only the 608 bytes it writes and then read "2.17 x requested" off the PMC counters)
        n_pel, n_vir, n_food, n_cells = res["counts"]
        extra = model_extra + A * cfg["num_agents"] * (8.0 * n_pel + 12.0 * n_vir + 12.0 * n_cells)
    kernel = None
    if with_obs: kernel = "k_step + k_grid_obs (persistent tensor: incremental clear)"
    if with_screen: kernel = "k_step + k_screen_obs"
    if with_ram: kernel = "k_step + k_ram_obs"
    return float(extra), float(model_extra), kernel


def roofline_block(res, A, K, ticks, cfg, workload, extra_bytes=0.0, kernel=None, model_extra=None):
    """roofline object for one measured run (see requested_bytes)."""
    n_pel, n_vir, n_food, n_cells = res["counts"]
    P, na = res["players"], cfg["num_agents"]
    req = requested_bytes(res["work"], res["counts"], P, na, ticks, res["pellet_cap"]) / K + extra_bytes
    t = res["kernel_ms"] * 1e-3
    # SURVEY 8(d) streaming model (every live entity once per tick): 8 N_p + 12 N_v + 72 N_c + 40 N_f + 112 P + 24 A
    b_tick = 8 * n_pel + 12 * n_vir + 72 * n_cells + 40 * n_food + 112 * P + 24 * na
    model = b_tick * A * ticks + (extra_bytes if model_extra is None else model_extra)
    traffic = tag = issue = None
    try:  # HBM bytes per step from the PMC counters, recorded separately by scripts/profile_round.sh on THIS kernel source
        tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
        ent = tj["runs"].get("%s@%d" % (workload, A))
        if ent and tj.get("source_sha") == source_sha() and res.get("sub_batches", 1) == 1 and res.get("world", 1) == 1:
            traffic, tag = ent["traffic_bytes_per_step"], "%s, source %s" % (tj.get("recorded", "?"), tj["source_sha"])
            issue = ent.get("issue")   # scripts/collect_profiles.py: the dominant kernel's SQ counters condensed (see "issue" below)
    except Exception:
        pass
    moved = traffic if traffic else req
    frac = moved / t / 1e9 / HBM_PEAK_GBS
    out = {"bound": "hbm", "achieved": moved / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
           # the two fractions, named for what they are:
           #   frac_hbm_traffic     = bytes that really left / entered HBM (PMC) / time / peak -- what the hardware did; == frac when PMC data
           #                          of this kernel source is committed, else None (frac then rests on the requested bytes)
           #   frac_streaming_model = SURVEY 8(d)'s B_tick model (every live entity once per tick) / time / peak -- NOT a bound for this
           #                          engine: it exceeds 1 where the engine does not move the model's bytes
           "frac_hbm_traffic": (traffic / t / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "frac_streaming_model": model / t / 1e9 / HBM_PEAK_GBS,
           "regime": ("bandwidth: the step's time is its bytes" if frac >= 0.30 else
                      "latency: launch floor, dependent instruction chains and serial pellet passes set the step's time, not its bytes"),
           "traffic": traffic, "traffic_source": tag, "requested_bytes_per_step": req, "algorithmic_bytes_per_step": req,
           # how much of what leaves HBM the kernels asked for (1 = no wasted re-reads / partial lines / spills); None without PMC data
           "frac_of_requested": (req / traffic) if traffic else None, "kernel_ms": res["kernel_ms"],
           "kernel": kernel or ("k_fused (one launch per env step)" if res["fused"] and P == 1 else ("k_quiet + k_step" if P == 1 else "k_step")),
           "arenas": A, "streaming_model_bytes_per_step": model, "streaming_model_bytes_per_arena_tick": b_tick,
           "model_speedup": model / t / 1e9 / HBM_PEAK_GBS,
           "work_per_step": {"front_finished_arena_steps": float(res["work"][0]) / K, "general_engine_arena_steps": float(res["work"][1]) / K,
                             "pellet_array_transfers": float(res["work"][2]) / K}}
    # The roofline that binds the general engine is not HBM but instruction issue and the dependent chain of the slowest arena.  From one SQ
    # pass of rocprofv3 over this workload on this kernel source (profiles/, sha-gated like `traffic`):
    #   frac_valu_issue     = SQ_INSTS_VALU x 2 cycles / (kernel time x measured clock x 1024 SIMDs): share of the VALU issue slots used
    #   mean_wave_residency = mean time a wavefront is resident / the launch's duration: 1 - this = wave slots standing empty while the launch
    #                         waits for its slowest arenas
    #   clock_ghz           = SQ_BUSY_CYCLES / 32 / kernel time: the shader clock under this load (nominal 2.4 GHz)
    if issue:
        out["frac_valu_issue"] = issue.get("frac_valu_issue"); out["mean_wave_residency"] = issue.get("mean_wave_residency")
        out["clock_ghz_measured"] = issue.get("clock_ghz"); out["issue_kernel"] = issue.get("kernel")
    else:
        out["frac_valu_issue"] = out["mean_wave_residency"] = out["clock_ghz_measured"] = None
    if res.get("sub_batches", 1) > 1:
        out["sub_batches"] = res["sub_batches"]; out["sub_batches_concurrent"] = res.get("concurrent")
    return out


def compact(rf, value, ms, cpu=None, cpu_cores=None):
    """one row of roofline.by_workload: [ms_per_step, env_steps_per_s, frac_hbm_traffic, traffic / requested, frac_valu_issue,
    mean_wave_residency, cpu_reference_env_steps_per_s, cpu_reference_cores]"""
    tr = rf.get("traffic")
    return [ms, value, rf.get("frac_hbm_traffic"), (tr / rf["requested_bytes_per_step"]) if tr else None, rf.get("frac_valu_issue"),
            rf.get("mean_wave_residency"), cpu, cpu_cores]


BY_WORKLOAD_COLUMNS = ["ms_per_step", "env_steps_per_s", "frac_hbm_traffic", "traffic_over_requested", "frac_valu_issue", "mean_wave_residency",
                       "cpu_reference_env_steps_per_s", "cpu_reference_cores"]


LINE_LIMIT = 6000          # the driver keeps an 8 KB tail of stdout + stderr: the ONE line stays well inside it (round 5's 52 KB line was lost)
FULL_RECORD = os.path.join(ROOT, "bench_full.json")   # everything else measured in the run (full roofline blocks, per-rank records, the vector surface)


def sig(x, n=4):
    """numbers of the printed line carry n significant digits (the full record keeps them all)"""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return x


ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "arenas", "algorithmic_bytes_per_step", "requested_bytes_per_step",
             "frac_of_requested", "frac_hbm_traffic", "frac_streaming_model", "streaming_model_bytes_per_step", "frac_valu_issue", "mean_wave_residency",
             "clock_ghz_measured", "measured_copy_GBs", "frac_of_measured_copy", "guide_copy_GBs", "frac_of_guide_copy", "traffic_source", "sub_batches")
CPU_KEEP = ("value", "unit", "cores", "kind", "sample", "c3m6_value", "c1_ticks_per_s_1core", "ticks_per_s_1core")


def compact_line(out):
    """The ONE line rank 0 prints: the contract's fields exactly as measured, `roofline` (the headline kernel's fields + one compact row per
    other workload of the run) and `cpu_baseline`; strings <= 120 characters, side figures at 4 significant digits.  Everything is in
    bench_full.json."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfgd = dict(out["config"]); cfgd["workload"] = cfgd["workload"][:120]; cfgd["parallelism"] = cfgd["parallelism"][:120]
    line["config"] = cfgd
    full_roof = out["roofline"]
    roof = {k: sig(full_roof[k]) for k in ROOF_KEEP if k in full_roof}
    roof["kernel"] = str(roof.get("kernel", ""))[:120]
    by = full_roof.get("by_workload")
    if by:
        # flat copies of the rows the round's targets are stated on (should a reader keep only the scalars of `roofline`)
        for key, name in (("C3m6@%d" % full_roof["arenas"], "C3m6"), ("C3m6@%d/pipe4" % full_roof["arenas"], "C3m6_pipe4"), ("mid@%d" % full_roof["arenas"], "mid"),
                          ("C1@%d" % full_roof["arenas"], "C1"), ("C5@%d" % full_roof["arenas"], "C5"), ("C5s@%d" % full_roof["arenas"], "C5s"),
                          ("task3@%d" % full_roof["arenas"], "task3"), ("task5@%d" % full_roof["arenas"], "task5"), ("task6@%d" % full_roof["arenas"], "task6"),
                          ("task5@%d/pipe4" % full_roof["arenas"], "task5_pipe4"), ("task6@%d/pipe4" % full_roof["arenas"], "task6_pipe4"), ("task10@%d" % full_roof["arenas"], "task10"),
                          ("Tick/30@%d" % full_roof["arenas"], "tick30"), ("C2@65536", "C2_65536"), ("C3m6@32768", "C3m6_32768")):
            row = by.get(key)
            if isinstance(row, list):
                roof["ms_" + name] = sig(row[0])
        roof["by_workload_columns"] = full_roof["by_workload_columns"]
        roof["by_workload"] = {k: (sig(v) if isinstance(v, list) else {"error": str(v.get("error"))[:80]}) for k, v in by.items()}
    gv = full_roof.get("gym_vector")
    if isinstance(gv, dict) and "error" not in gv:
        # the RL surface: [host us per step, us per step] un-pipelined, then with the sub-batching AgarioVectorEnv picks by itself
        roof["gym_vector_us"] = {k: sig([v["host_us_per_step"], v["us_per_step"]], 3) for k, v in gv.items()}
        ga = full_roof.get("gym_vector_auto")
        if isinstance(ga, dict) and "error" not in ga:
            roof["gym_vector_auto_us"] = {k: sig([v["host_us_per_step"], v["us_per_step"]], 3) for k, v in ga.items()}
    line["roofline"] = roof
    if out.get("cpu_baseline"):
        cb = {k: sig(out["cpu_baseline"][k], 5) for k in CPU_KEEP if k in out["cpu_baseline"]}
        cb["sample"] = cb["sample"][:120]
        line["cpu_baseline"] = cb
    line["capacity_flags_raised"] = out.get("capacity_flags_raised")
    if out.get("world_size", 1) > 1:
        line["world_size"] = out["world_size"]; line["backend"] = out["backend"]; line["result_gather"] = out["result_gather"]; line["obs_gather_ran"] = out["obs_gather_ran"]
        line["rank_devices"] = [str(d)[-24:] for d in out["rank_devices"]]
        # per rank: [ms per step on its own clock, ms per step by HIP events, ms spent waiting for collectives in total, result collectives, obs collectives]
        line["ranks_columns"] = ["ms_per_step", "kernel_ms_per_step", "gather_wait_ms_total", "result_collectives", "obs_collectives", "requested_GBs"]
        line["ranks"] = [sig([r["ms_per_step"], r["kernel_ms_per_step"], r["gather_wait_ms_total"], r["result_collectives"], r["obs_collectives"], r.get("requested_GBs")]) for r in out["ranks"]]
    line["full_record"] = os.path.basename(FULL_RECORD)
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > LINE_LIMIT and "by_workload" in roof:       # never again a line the driver cannot keep: rows go first, the contract fields never
        for k in [k for k in roof["by_workload"] if "/pipe" in k or k.startswith("Tick/") or k.startswith("task")]:
            del roof["by_workload"][k]
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt) <= LINE_LIMIT:
                break
    if len(txt) > LINE_LIMIT:
        for k in ("by_workload", "by_workload_columns", "gym_vector_us", "gym_vector_auto_us"):
            roof.pop(k, None)
        txt = json.dumps(line, separators=(",", ":"))
    return txt


def emit(out):
    """writes the full record beside bench.py and prints the one compact line (stdout carries nothing else that starts with '{')"""
    try:
        with open(FULL_RECORD, "w") as f:
            json.dump(out, f, indent=1, default=float)
    except OSError:
        pass
    print(compact_line(out))


def reference_rate(cfg, seconds=0.5, dt=1.0 / 30):
    """the reference engine (oracle/_ref, kind "reference"; else the C port) on ONE host core on this env configuration, random policy:
    engine ticks per second over a bounded sample"""
    kw = dict(num_agents=cfg["num_agents"], ticks_per_step=cfg["ticks_per_step"], arena_size=cfg["arena_size"], pellet_regen=cfg["pellet_regen"],
              num_pellets=cfg["num_pellets"], num_viruses=cfg["num_viruses"], num_bots=cfg["num_bots"], reward_type=cfg["reward_type"], c_death=cfg["c_death"],
              mode=cfg["mode_number"], dt=cfg.get("dt", dt))
    try:
        from oracle import refbind as B
        if not B.available():
            raise ImportError
        env, kind = B.RefEnv(**kw), "reference"
    except Exception:
        from oracle import orabind as B
        if not B.available():
            B.build()
        env, kind = B.OraEnv(**kw), "port"
    if cfg.get("screen_respawn"):
        env.set_screen_hook(True)
    env.seed(42); env.reset(True)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n += env.run_random(400, policy_seed=n + 1, allow_actions=True)
    rate = n / (time.perf_counter() - t0)
    env.close()
    return rate, kind


def vector_surface(torch, A, K, Wm, dev_index, sub_batches=1):
    """The RL surface itself (agarcl_amd/vector_env.py AgarioVectorEnv = gym.make's N > 1 case, one agarcl_vec_step per step): gym "normal"
    preset (= C2's arena), a fixed device action batch, K steps.  host_us_per_step: what step() costs the host (enqueue only, nothing
    waits); gym_vector_steps_per_s: arena-steps per second through the surface, GPU time included."""
    from agarcl_amd.vector_env import AgarioVectorEnv
    out = {}
    for tag, obs, kw in (("no_obs", "none", {}), ("ram_obs", "ram", {}), ("screen_obs_84", "screen", dict(screen_len=84))):
        venv = AgarioVectorEnv(A, obs_type=obs, device=dev_index, difficulty="normal", strict_flags=False, sub_batches=sub_batches, **kw)
        venv.reset(seed=10000)
        dev = venv.device
        move = torch.rand((A, 2), device=dev) * 2 - 1; kind = torch.zeros(A, dtype=torch.int32, device=dev)
        for _ in range(Wm):
            venv.step((move, kind))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            venv.step((move, kind))
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        out[tag] = {"host_us_per_step": host / K * 1e6, "us_per_step": total / K * 1e6, "gym_vector_steps_per_s": A * K / total,
                    "env_steps_per_s": A * K * venv.options["ticks_per_step"] / total}
        venv.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--arenas", type=int, default=ARENAS_PER_GPU, help="arenas per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the %d-arena roofline_large run" % LARGE_ARENAS)
    ap.add_argument("--no-full", action="store_true", help="skip the other workloads of roofline.by_workload (full rule set, mid-game, C1, observations, pipelined forms, tasks, Tick/N, the vector surface)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS) + ["tick%d" % n for n in TICK_BOTS] + ["task%d" % m for m in range(1, 11)],
                    help="C2 = the headline metric's configuration; tickN = bench/main.cpp's Tick/N population (N ExampleBots, no Player), batched; "
                         "taskM = the reference's RL task M (bench/tasks_configs/mode_M.json) with its 128 x 128 agent-view frame every step")
    ap.add_argument("--sub-batches", type=int, default=1, help="the rank's arenas as this many independent sub-batches on HIP streams of their own (agarcl_pipe_*)")
    ap.add_argument("--gather", default="block", choices=["block", "step"], help="multi-GPU: how (reward, done) reaches rank 0")
    ap.add_argument("--gather-block", type=int, default=32, choices=[8, 16, 32], help="multi-GPU, --gather block: steps per collective")
    ap.add_argument("--gather-obs", default="none", choices=["none", "screen"], help="multi-GPU: also gather every step's uint8 frames")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))            # children first: no GPU call has been made in this process
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the number of rank processes must equal --gpus" % (args.gpus, world))
    if args.workload.startswith("tick"):      # the literal bench/main.cpp population: engine-level ticks, single GPU
        if world != 1:
            raise SystemExit("--workload tickN is a single-GPU measurement")
        nb = int(args.workload[4:])
        res, cfg = run_tick_workload(nb, args.arenas, args.steps, args.warmup)
        roof = roofline_block(res, args.arenas, args.steps, 4, cfg, args.workload, kernel="k_step (agarcl_tick: 4 engine ticks per launch)")
        out = {"metric": "env-steps/sec (arenas x ticks/s)", "value": args.arenas * 4 * args.steps / res["elapsed"], "unit": "env-steps/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "bench/main.cpp Tick/%d batched: %d arenas x %d ExampleBots (no Player), default engine 250x250 / 500 pellets / 10 viruses, dt 1/60 s, "
                                      "4 ticks per launch" % (nb, args.arenas, nb), "arenas_total": args.arenas, "ticks_per_step": 4, "parallelism": "single GPU"},
               "roofline": roof, "capacity_flags_raised": int((res["flags"] != 0).sum())}
        if not args.no_cpu_baseline:
            rate, kind = tick_reference_rate(nb, 2.0)
            out["cpu_baseline"] = {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": kind, "sample": "2 s of Engine::tick at dt 1/60 s on the same population, one core"}
        emit(out)
        return
    screen = (84, 84, False)
    if args.workload.startswith("task"):
        cfg, screen, desc = task_workload(int(args.workload[4:]))
        rand_act, with_obs, with_screen, with_ram, start_mass = True, False, True, False, 0
        desc = desc.replace("%%", "%")
    else:
        wl = dict(WORKLOADS[args.workload])
        desc, rand_act, with_obs, with_screen = wl.pop("desc"), wl.pop("rand_act", False), wl.pop("grid_obs", False), wl.pop("screen_obs", False)
        start_mass = wl.pop("start_mass", 0)
        with_ram = wl.pop("ram_obs", False)
        cfg = dict(CFG); cfg.update(wl)

    # more hardware queues than the runtime's default 4, so that sub-batch streams (agarcl_pipe_*) find queues of their own beside torch's
    # streams; must be in the environment before the first HIP call (read by the HIP runtime, nothing else)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    # one process per GPU.  (AGAR_BENCH_BACKEND=gloo + fewer GPUs than ranks is only for exercising the
    # multi-rank code path on a single-GPU box; the driver's runs use nccl == RCCL over xGMI.)
    backend = os.environ.get("AGAR_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and world > torch.cuda.device_count():
        # (device_count() does not initialise the GPU.)  RCCL refuses two ranks on one device deep inside init ("duplicate GPU"); say it here
        raise SystemExit("--gpus %d over nccl (RCCL) needs %d GPUs on this node, torch sees %d: one process per GPU "
                         "(AGAR_BENCH_BACKEND=gloo exercises the multi-rank path on fewer GPUs)" % (world, world, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from agarcl_amd import _capi as _c, build as _hip_build
    if not os.path.exists(_c.HIP_SO) and rank == 0:
        _hip_build.build()          # normally prebuilt by __graft_entry__.build(); never rebuilt when present
    if world > 1:
        dist.barrier()              # nobody loads the library while rank 0 may still be writing it
    from agarcl_amd.vec_env import VecEnvironment, default_sub_batches
    from agarcl_amd import dist as agdist

    A, K, Wm = args.arenas, args.steps, args.warmup
    ticks = cfg["ticks_per_step"]
    res = run_workload(torch, np, VecEnvironment, agdist, dev, dev_index, rank, world, A, K, Wm, cfg, rand_act, with_obs, with_screen,
                       args.gather, args.gather_obs, args.gather_block, start_mass, with_ram, sub_batches=args.sub_batches, screen=screen)
    devs = [None] * world   # which device every rank ran on: lets the driver see "RCCL saw N ranks on N GPUs"
    if world > 1:
        dist.all_gather_object(devs, "%s:%d" % (os.uname().nodename, dev_index))
    else:
        devs = ["%s:%d" % (os.uname().nodename, dev_index)]
    value = world * A * ticks * K / res["elapsed"]
    if rank == 0:
        extra, model_extra, kernel = obs_bytes(res, A, cfg, with_obs, with_screen, with_ram, screen)
        roof = roofline_block(res, A, K, ticks, cfg, args.workload, float(extra), kernel, float(model_extra))
        roof["note"] = ("achieved = HBM bytes one env step moves (PMC FETCH_SIZE x2 + WRITE_SIZE of the same kernel source when profiles/ "
                        "holds them -> `traffic`; otherwise the bytes the kernels request, counted by the kernels themselves) / the step's "
                        "HIP-event time; frac <= 1 by construction.  frac_streaming_model = SURVEY 8(d)'s streaming-model bytes over the same time: "
                        "it exceeds 1 because the engine does not stream (state stays in registers across the ticks of a step, pellets are "
                        "read only when a cell leaves its pellet-free disc): not a bound.  frac == frac_hbm_traffic when PMC data of this kernel "
                        "source is committed.  by_workload: every other workload measured in this run on the same clock, one row each "
                        "(by_workload_columns); '<w>/pipe%d' = the same arenas as %d independent sub-batches on streams of their own" % (PIPE_K, PIPE_K))
        if world > 1:
            par = "arena-sharded x%d; (reward, done) -> rank 0 %s%s%s" % (
                world, "in async blocks of %d steps" % args.gather_block if args.gather == "block" else "per step",
                "; + uint8 screen frames per step" if args.gather_obs == "screen" else "",
                "; %d sub-batches per rank" % args.sub_batches if args.sub_batches > 1 else "")
        else:
            par = "single GPU" + (", %d sub-batches on streams of their own" % args.sub_batches if args.sub_batches > 1 else "")
        out = {
            "metric": "env-steps/sec (arenas x ticks/s)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": res["elapsed"] / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc % ((A, A) if desc.count("%d") == 2 else A), "arenas_total": world * A, "ticks_per_step": ticks, "parallelism": par},
            "gym_steps_per_s": value / ticks,
            "world_size": world, "backend": (backend if world > 1 else None), "rank_devices": devs,
            # one entry per rank: its own wall / HIP-event step time, host time spent waiting for collectives, collectives and bytes it sent
            "ranks": res["ranks"], "result_gather": (args.gather if world > 1 else None), "obs_gather_ran": bool(world > 1 and args.gather_obs != "none"),
            "roofline": roof,
            "capacity_flags_raised": int((res["flags"] != 0).sum()),
        }
        headline = args.workload == "C2" and world == 1 and args.sub_batches == 1
        by = {}                      # roofline.by_workload: one compact row per other workload measured in this run (BY_WORKLOAD_COLUMNS)
        detail = {}                  # the full roofline blocks of the same runs (top-level `roofline_full`, as in earlier rounds)
        cpu = cpu_baseline() if (headline and not args.no_cpu_baseline) else None
        ncores = cpu["cores"] if cpu else None

        def measure(name, wcfg, a_, k_, w_, ra=False, wo=False, ws=False, wr=False, sm=0, sub=1, scr=(84, 84, False), cpu_rate=None, cpu_cores=None, label=None):
            key = "%s@%d%s" % (name, a_, "/pipe%d" % sub if sub > 1 else "")
            try:
                r2 = run_workload(torch, np, VecEnvironment, agdist, dev, dev_index, 0, 1, a_, k_, w_, wcfg, ra, wo, ws, "block", "none", 32, sm, wr, sub_batches=sub, screen=scr)
                ex, mex, kern = obs_bytes(r2, a_, wcfg, wo, ws, wr, scr)
                rf = roofline_block(r2, a_, k_, wcfg["ticks_per_step"], wcfg, name, ex, kern, mex)
                v_ = a_ * wcfg["ticks_per_step"] * k_ / r2["elapsed"]; ms_ = r2["elapsed"] / k_ * 1e3
                rf["value_env_steps_per_s"] = v_; rf["ms_per_step"] = ms_
                rf["mean_counts_pellets_viruses_foods_cells"] = [float(x) for x in r2["counts"]]
                if label:
                    rf["workload"] = label
                by[key] = compact(rf, v_, ms_, cpu_rate, cpu_cores)
                detail[key] = rf
                return rf
            except Exception as ex:  # the headline line must not depend on it
                by[key] = {"error": str(ex)}; detail[key] = {"error": str(ex)}
                return None

        def wl_args(name):
            w2 = dict(WORKLOADS[name]); lab = w2.pop("desc")
            kw = dict(ra=w2.pop("rand_act", False), sm=w2.pop("start_mass", 0), wo=w2.pop("grid_obs", False), ws=w2.pop("screen_obs", False), wr=w2.pop("ram_obs", False))
            c2 = dict(CFG); c2.update(w2)
            return c2, kw, lab

        if headline and not args.no_large:
            # the same kernels where they are bandwidth- rather than latency-bound: >= 50 k arenas (north star), short runs
            c2cpu = (cpu["value"], ncores) if cpu else (None, None)
            by["C2@%d" % A] = compact(roof, value, out["ms_per_step"], *c2cpu)
            for a_, k_, w_, top in ((16384, 100, 20, None), (LARGE_ARENAS, 200, 40, "roofline_large"), (131072, 100, 20, None), (XLARGE_ARENAS, 100, 20, "roofline_xlarge")):
                rf = measure("C2", cfg, a_, k_, w_, ra=rand_act, cpu_rate=c2cpu[0], cpu_cores=c2cpu[1])
                if top:
                    out[top] = rf if rf is not None else {"error": by["C2@%d" % a_].get("error")}
            out["roofline_sweep"] = [{"arenas": a_, "ms_per_step": detail["C2@%d" % a_].get("ms_per_step"), "value_env_steps_per_s": detail["C2@%d" % a_].get("value_env_steps_per_s"),
                                      "achieved": detail["C2@%d" % a_].get("achieved"), "frac": detail["C2@%d" % a_].get("frac"), "unit": "GB/s"} for a_ in SWEEP_ARENAS]
        if headline and not args.no_full:
            # every other BASELINE config on the same clock: the full rule set at mass 1000 (configs[2]), a learning agent's mid-game, configs[0]
            # (the bench/main.cpp population, batched), configs[4] (grid / screen observation on top of the full rule set) -- each also as
            # PIPE_K independent sub-batches --, and the full rule set at 32768 arenas on one GPU (configs[3]'s per-node size / 8)
            c6cpu = (cpu["c3m6_value"], ncores) if cpu else (None, None)
            c1cpu = (cpu["c1_ticks_per_s_1core"], 1) if cpu else (None, None)
            for name, fk, fw, cr in (("C3m6", 100, 20, c6cpu), ("mid", 150, 400, (None, None)), ("C1", 100, 20, c1cpu), ("C5", 60, 20, c6cpu), ("C5s", 60, 20, c6cpu)):
                c2, kw, lab = wl_args(name)
                rf = measure(name, c2, A, fk, fw, cpu_rate=cr[0], cpu_cores=cr[1], label=lab % A, **kw)
                if name == "C1" and rf is not None and cpu:
                    rf["cpu_reference_ticks_per_s_1core"] = cpu["c1_ticks_per_s_1core"]
                # ... and as PIPE_K independent sub-batches (GPU_MAX_HW_QUEUES was raised above: that many queues beside torch's streams)
                measure(name, c2, A, fk, fw, sub=PIPE_K, cpu_rate=cr[0], cpu_cores=cr[1], label=(lab % A) + " -- as %d sub-batches" % PIPE_K, **kw)
            c2, kw, lab = wl_args("C3m6")
            measure("C3m6", c2, 32768, 40, 10, cpu_rate=c6cpu[0], cpu_cores=c6cpu[1], label=lab % 32768, **kw)
        if world == 1:  # measured roofline next to the nominal one: device stream copy and fill of 1 GiB
            try:
                src = torch.empty(1 << 28, dtype=torch.int32, device=dev); dst = torch.empty_like(src)
                def _bw(fn, nbytes):
                    for _ in range(2):
                        fn()
                    torch.cuda.synchronize(); t = time.perf_counter()
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                    return nbytes * 5 / (time.perf_counter() - t) / 1e9
                copy_gbs = _bw(lambda: dst.copy_(src), 2 * src.numel() * 4)
                fill_gbs = _bw(lambda: dst.fill_(1), src.numel() * 4)
                del src, dst
                for r in [roof] + [v for v in detail.values() if v and "achieved" in v]:
                    r["measured_copy_GBs"] = copy_gbs; r["measured_fill_GBs"] = fill_gbs
                    r["frac_of_measured_copy"] = r["achieved"] / copy_gbs
                    r["guide_copy_GBs"] = GUIDE_COPY_GBS; r["frac_of_guide_copy"] = r["achieved"] / GUIDE_COPY_GBS
            except Exception:
                pass
        if headline and not args.no_full:
            # bench/main.cpp:14-38 literally: Tick/{0,5,10,20,30} -- N ExampleBots and no Player on the default engine -- batched at this arena
            # count, with the reference engine's own rate on one host core beside each
            table = {}
            for nb in TICK_BOTS:
                key = "Tick/%d@%d" % (nb, A)
                try:
                    r3, c3 = run_tick_workload(nb, A, 50, 10, dev_index)
                    ent = {"gpu_env_steps_per_s": A * 4 * 50 / r3["elapsed"], "gpu_us_per_4_ticks": r3["kernel_ms"] * 1e3}
                    if not args.no_cpu_baseline:
                        ent["cpu_ticks_per_s_1core"], ent["cpu_kind"] = tick_reference_rate(nb)
                    table[key] = ent
                    rf = roofline_block(r3, A, 50, 4, c3, "tick%d" % nb, kernel="k_step (agarcl_tick)")
                    by[key] = compact(rf, ent["gpu_env_steps_per_s"], r3["elapsed"] / 50 * 1e3, ent.get("cpu_ticks_per_s_1core"), 1 if "cpu_ticks_per_s_1core" in ent else None)
                except Exception as ex:
                    table[key] = {"error": str(ex)}; by[key] = {"error": str(ex)}
            out["bench_main_cpp_tick"] = table
            # the reference's ten RL tasks (bench/tasks_configs/mode_{1..10}.json), each with its 128 x 128 agent-view frame written every step
            tasks = {}
            for m in sorted(paper_tasks()):
                tcfg, tscr, tdesc = task_workload(m)
                cr = (None, None)
                if not args.no_cpu_baseline:
                    try:
                        cr = (reference_rate(tcfg, 0.4)[0], 1)
                    except Exception:
                        pass
                rf = measure("task%d" % m, tcfg, A, 40, 10, ra=True, ws=True, scr=tscr, cpu_rate=cr[0], cpu_cores=cr[1], label=tdesc.replace("%%", "%") % (A, A))
                if rf is not None:
                    tasks["task%d" % m] = by["task%d@%d" % (m, A)]
                # ... and as free-running ranges (vec_env.default_sub_batches: 4 where the general engine does the work -- modes 5 / 6, bots --: one
                # range's frame kernel runs under another's step; what the recv / send halves of AgarioVectorEnv(halves=True) and
                # PipelinedVecEnvironment give, NOT the full-batch step(), which stays one range: scripts/gpu_vec_pipe_ab.py)
                dsub = default_sub_batches(A, tcfg["num_agents"], tcfg["num_bots"], tcfg["mode_number"])
                if dsub > 1:
                    measure("task%d" % m, tcfg, A, 40, 10, ra=True, ws=True, scr=tscr, sub=dsub, cpu_rate=cr[0], cpu_cores=cr[1],
                            label=tdesc.replace("%%", "%") % (A, A) + " -- as %d free-running sub-batches" % dsub)
            roof["tasks"] = {"columns": BY_WORKLOAD_COLUMNS, "rows": tasks,
                             "what": "the reference's bench/tasks_configs/mode_{1..10}.json at %d arenas: k_step (+ k_quiet / k_fused) + k_screen_obs 128x128x4 per step" % A}
            try:   # the RL surface itself: AgarioVectorEnv.step = ONE agarcl_vec_step (step + bookkeeping / auto-reset + observation)
                vs = vector_surface(torch, A, 200, 30, dev_index)
                roof["gym_vector"] = vs
                out["gym_vector_steps_per_s"] = vs["no_obs"]["gym_vector_steps_per_s"]
                roof["gym_vector_auto"] = vector_surface(torch, A, 200, 30, dev_index, sub_batches="auto")   # the surface's own choice of sub-batching
            except Exception as ex:
                roof["gym_vector"] = {"error": str(ex)}
        if headline and not args.no_full:
            out["roofline_full"] = {k: v for k, v in detail.items() if not k.startswith("C2@")}
        for r in [v for v in detail.values() if v and "achieved" in v and "guide_copy_GBs" not in v]:   # (blocks measured after the copy was timed)
            if "measured_copy_GBs" in roof:
                r["measured_copy_GBs"] = roof["measured_copy_GBs"]; r["measured_fill_GBs"] = roof["measured_fill_GBs"]; r["frac_of_measured_copy"] = r["achieved"] / roof["measured_copy_GBs"]
            r["guide_copy_GBs"] = GUIDE_COPY_GBS; r["frac_of_guide_copy"] = r["achieved"] / GUIDE_COPY_GBS
        if by:
            roof["by_workload_columns"] = BY_WORKLOAD_COLUMNS
            roof["by_workload"] = by
        if cpu:
            out["cpu_baseline"] = cpu
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
