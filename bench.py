#!/usr/bin/env python
"""Headline benchmark: env-steps/s (arenas x engine ticks per second) of the batched HIP engine.

Workload = BASELINE.json configs[1] / SURVEY.md 8(d) "C2": 4096 arenas per GPU, each 1000x1000,
1000 pellets, 0 viruses, 1 agent, no bots, pellet regen, mode 0, dt = 1/30, 4 ticks per env step,
action "none", (dx,dy) ~ U(-1,1)^2 pre-generated in HBM, arena seeds = 10000 + global arena index.
A bench "step" is one agarcl_step launch = 4 engine ticks of every arena.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the roofline / cpu_baseline fields).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARENAS_PER_GPU = 4096
CFG = dict(num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000, num_viruses=0,
           num_bots=0, reward_type=1, c_death=0, mode_number=0)
# The headline line is C2.  The other SURVEY 8(d) workloads are selectable for DESIGN.md's measurement table only.
WORKLOADS = {
    "C2": dict(desc="C2: %d arenas/GPU x 1 agent, 1000x1000 arena, 1000 pellets, 0 viruses, mode 0, 4 ticks/step, random (dx,dy), action none"),
    "C3m0": dict(num_viruses=25, rand_act=True,
                 desc="C3/mode 0: %d arenas/GPU x 1 agent, 1000x1000, 1000 pellets, 25 viruses, mode 0, 4 ticks/step, random (dx,dy), action ~ U{0,1,2}"),
    "C3m6": dict(num_viruses=25, mode_number=6, rand_act=True,
                 desc="C3/mode 6: %d arenas/GPU x 1 agent (mass 1000), 1000x1000, 1000 pellets, 25 viruses, mode 6, 4 ticks/step, random (dx,dy), action ~ U{0,1,2}"),
    "C5": dict(num_viruses=25, mode_number=6, rand_act=True, grid_obs=True,
               desc="C5: C3/mode 6 + int32 grid observation [%d][8][128][128] written once per step"),
    "C5s": dict(num_viruses=25, mode_number=6, rand_act=True, screen_obs=True,
                desc="C5 (screen): C3/mode 6 + uint8 screen observation [%d][84][84][3] written once per step"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)


def cpu_baseline(seconds_budget=12.0):
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
        mk = lambda: B.RefEnv(num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                              num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode=0)
        kind = "reference"
    else:
        from oracle import orabind as B
        if not B.available():
            B.build()
        mk = lambda: B.OraEnv(num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                              num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode=0)
        kind = "port"
    # calibrate on one core
    e = mk(); e.seed(10000); e.reset(True)
    t0 = time.perf_counter(); e.run_random(4000, policy_seed=1, allow_actions=False); dt1 = time.perf_counter() - t0
    rate1 = 4000 / dt1
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
    return {"value": total / el, "unit": "env-steps/s", "cores": cores, "kind": kind,
            "sample": "%.0f s of the C2 workload (1000x1000, 1000 pellets, 1 agent, random dx/dy, action none), one engine "
                      "per host thread, %d threads, %d arena-ticks in total; single-thread rate %.0f ticks/s"
                      % (seconds_budget, cores, total, rate1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--arenas", type=int, default=ARENAS_PER_GPU, help="arenas per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS), help="C2 = the headline metric's configuration")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    desc, rand_act, with_obs, with_screen = wl.pop("desc"), wl.pop("rand_act", False), wl.pop("grid_obs", False), wl.pop("screen_obs", False)
    CFG.update(wl)

    import torch
    import numpy as np
    from agarcl_amd import _capi as _c, build as _hip_build
    if not os.path.exists(_c.HIP_SO) and int(os.environ.get("RANK", "0")) == 0:
        _hip_build.build()          # normally prebuilt by __graft_entry__.build(); never rebuilt when present
    from agarcl_amd.vec_env import VecEnvironment
    from agarcl_amd import dist as agdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    # one process per GPU.  (AGAR_BENCH_BACKEND=gloo + fewer GPUs than ranks is only for exercising the
    # multi-rank code path on a single-GPU box; the driver's runs use nccl == RCCL over xGMI.)
    backend = os.environ.get("AGAR_BENCH_BACKEND", "nccl")
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

    A = args.arenas
    K, Wm = args.steps, args.warmup
    lo, hi = rank * A, (rank + 1) * A  # weak scaling: every GPU owns `A` arenas
    env = VecEnvironment(A, device=dev_index, **CFG)
    env.seed(agdist.arena_seeds(10000, lo, hi))
    env.reset(reset_ids=True)

    # synthetic random policy, resident in HBM before the timed region: counter-based per (arena, step)
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    dxdy = (torch.rand((K + Wm, A, 1, 2), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    act = torch.zeros((K + Wm, A, 1), dtype=torch.int32, device=dev)
    if rand_act:
        act = torch.randint(0, 3, (K + Wm, A, 1), generator=g, device=dev, dtype=torch.int32)
    obs = torch.empty((A, 8, 128, 128), dtype=torch.int32, device=dev) if with_obs else None
    scr = torch.empty((A, 84, 84, 3), dtype=torch.uint8, device=dev) if with_screen else None
    # Multi-GPU: (reward, done) of every step reaches rank 0 through RCCL gathers of 8 steps at a time (one contiguous
    # block of the engine's 16-slot result ring, zero copy), asynchronous and double-buffered by ring half: a per-step
    # collective (~20 us of latency) would cost more than the 11 us step itself.
    BLK = 8
    gather = agdist.ResultGatherer(BLK * A, dev) if world > 1 else None
    eng = env.engine

    def one_step(k):
        eng.set_actions_device(dxdy[k].data_ptr(), act[k].data_ptr())
        if gather is not None:
            nxt = (eng.last_slot() + 1) % (2 * BLK)
            if nxt % BLK == 0:
                gather.wait_slot(nxt // BLK)    # the engine is about to overwrite this half of the ring
        eng.step(CFG["ticks_per_step"])
        if obs is not None:
            eng.grid_obs(128, True, True, True, True, out_ptr=obs.data_ptr())
        if scr is not None:
            eng.screen_obs(84, 84, out_ptr=scr.data_ptr())
        if gather is not None:                  # RCCL gather of 8 steps of (reward, done) straight from engine memory
            s_ = eng.last_slot()
            if s_ % BLK == BLK - 1:
                h_ = s_ // BLK
                gather.gather_packed(h_, env.packed_ring[h_ * BLK:(h_ + 1) * BLK].reshape(-1, 2))

    def flush():                                # results of a partial last block still go to rank 0
        s_ = eng.last_slot()
        if gather is not None and s_ % BLK != BLK - 1:
            h_ = s_ // BLK
            gather.wait_slot(h_)
            gather.gather_packed(h_, env.packed_ring[h_ * BLK:(h_ + 1) * BLK].reshape(-1, 2))

    for k in range(Wm):
        one_step(k)
    flush()
    if gather is not None:
        gather.wait_all()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(Wm, Wm + K):
        one_step(k)
    flush()
    ev1.record()
    if gather is not None:
        gather.wait_all()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / K  # HIP events on the launch stream: avg per launch incl. gaps
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    flags = eng.flags()
    counts = eng.counts().astype(np.float64).mean(axis=0)  # pellets, viruses, foods, cells per arena
    ticks = CFG["ticks_per_step"]
    value = world * A * ticks * K / elapsed
    if rank == 0:
        # algorithmic bytes per arena-tick, SURVEY.md 8(d): 8 N_p + 12 N_v + 72 N_c + 40 N_f + 112 P + 24 A
        b_tick = 8 * counts[0] + 12 * counts[1] + 72 * counts[3] + 40 * counts[2] + 112 * 1 + 24 * 1
        bytes_per_launch = b_tick * A * ticks
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the PMC counters (collected separately: profiles/r01_pmc_traffic.json)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if A == 4096 and ticks == 4 and args.workload == "C2":
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec (arenas x ticks/s)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc % A,
                       "arenas_total": world * A, "ticks_per_step": ticks,
                       "parallelism": "arena-sharded x%d, every step's (reward, done) gathered to rank 0 in asynchronous blocks of 8 steps" % world if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel": "k_fused (one launch per env step)" if args.workload in ("C2", "C3m0") else "k_quiet + k_step (the two launches of one env step)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_arena_tick": b_tick,
                         "moved_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "note": "achieved = SURVEY 8(d) streaming-model bytes / time; frac > 1 is possible because the engine does "
                                 "not stream: state stays in registers across the 4 ticks and pellets are read only when a cell "
                                 "leaves its pellet-free disc (traffic = bytes really moved, PMC).  The step is latency/issue "
                                 "bound, not HBM bound: see DESIGN.md section 5"},
            "capacity_flags_raised": int((flags != 0).sum()),
        }
        if with_screen:
            out["roofline"]["algorithmic_bytes_per_launch"] = bytes_per_launch + A * 84 * 84 * 3
            out["roofline"]["achieved"] = out["roofline"]["algorithmic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
            out["roofline"]["kernel"] = "k_quiet + k_step + k_screen_obs"
        if with_obs:
            out["roofline"]["algorithmic_bytes_per_launch"] = bytes_per_launch + A * 8 * 128 * 128 * 4
            out["roofline"]["achieved"] = out["roofline"]["algorithmic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
            out["roofline"]["kernel"] = "k_step + k_grid_obs"
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
                out["roofline"]["measured_copy_GBs"] = _bw(lambda: dst.copy_(src), 2 * src.numel() * 4)
                out["roofline"]["measured_fill_GBs"] = _bw(lambda: dst.fill_(1), src.numel() * 4)
                del src, dst
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline and args.workload == "C2":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
