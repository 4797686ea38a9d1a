"""Probe: k independent sub-batches of one 4096-arena job, each on its own HIP stream (arenas never interact, so a sub-batch need not wait for
another's slowest arena), against the lock-step launch.  Prints us per 4096 arena-steps for k = 1, 2, 4, 8 and a few workloads.
    python scripts/gpu_pipe_probe.py [--workloads C3m6,C5s,C1,mid] [--ks 1,2,4,8] [--arenas 4096] [--steps 100]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from agarcl_amd.vec_env import VecEnvironment

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="C3m6,C5s,C5,C1,mid"); ap.add_argument("--ks", default="1,2,4,8")
ap.add_argument("--stagger-us", type=float, default=400.0)
ap.add_argument("--stagger", type=float, default=0.0, help="fraction of a step by which sub-batch j lags sub-batch j-1 at the start (GPU-side sleep)")
ap.add_argument("--own", type=int, default=1, help="1: engine-owned HIP streams, 0: torch pool streams")
ap.add_argument("--arenas", type=int, default=4096); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--warmup", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
out = open(os.path.join(ROOT, "gpurun_out", "pipe_probe.log"), "a") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
def say(s):
    print(s, flush=True)
    if out: out.write(s + "\n"); out.flush()

for name in a.workloads.split(","):
    wl = dict(bench.WORKLOADS[name]); wl.pop("desc"); rand_act = wl.pop("rand_act", False)
    with_obs, with_screen, with_ram = wl.pop("grid_obs", False), wl.pop("screen_obs", False), wl.pop("ram_obs", False)
    start_mass = wl.pop("start_mass", 0)
    cfg = dict(bench.CFG); cfg.update(wl)
    for k in [int(x) for x in a.ks.split(",")]:
        A = a.arenas; n = A // k
        streams = [torch.cuda.Stream(dev) for _ in range(k)]
        envs, obs = [], []
        K, W = a.steps, (400 if name == "mid" else a.warmup)
        g = torch.Generator(device=dev); g.manual_seed(1234)
        dxdy = (torch.rand((K + W, A, 1, 2), generator=g, device=dev) * 2 - 1).contiguous()
        act = torch.randint(0, 3, (K + W, A, 1), generator=g, device=dev, dtype=torch.int32) if rand_act else torch.zeros((K + W, A, 1), dtype=torch.int32, device=dev)
        for j in range(k):
            with torch.cuda.stream(streams[j]):
                e = VecEnvironment(n, device=0, strict_flags=False, use_torch_stream=not a.own, **cfg)
                import numpy as np
                e.seed(np.arange(10000 + j * n, 10000 + (j + 1) * n, dtype=np.uint32)); e.reset(reset_ids=True)
                if start_mass:
                    from agarcl_amd import snapshot
                    sn = snapshot.save_arena(e.engine, 0, cfg)
                    for pl in sn["players"]:
                        for cell in pl["cells"]: cell["mass"] = int(start_mass)
                    for q in range(n):
                        sn["seed"] = 10000 + j * n + q; snapshot.load_arena(e.engine, q, sn, reset_ids=True)
                envs.append(e)
                obs.append(torch.empty((n, 8, 128, 128), dtype=torch.int32, device=dev) if with_obs else (torch.empty((n, 84, 84, 3), dtype=torch.uint8, device=dev) if with_screen else None))
        torch.cuda.synchronize()
        ptr = [[(dxdy[s, j * n:(j + 1) * n].data_ptr(), act[s, j * n:(j + 1) * n].data_ptr()) for j in range(k)] for s in range(K + W)]
        def step(s):
            for j in range(k):
                eng = envs[j].engine
                eng.step_actions(ptr[s][j][0], ptr[s][j][1], 4)
                if with_obs: eng.grid_obs(128, True, True, True, True, out_ptr=obs[j].data_ptr(), persistent=True)
                if with_screen: eng.screen_obs(84, 84, out_ptr=obs[j].data_ptr())
        for s in range(W): step(s)
        for e in envs: e.sync()
        torch.cuda.synchronize()
        if a.stagger > 0 and k > 1:     # sub-batch j starts j * stagger * (one lock-step step) late: its step then runs under its predecessor's observation kernel
            for j in range(1, k):
                with torch.cuda.stream(envs[j].torch_stream()):
                    torch.cuda._sleep(int(a.stagger * j * a.stagger_us * 2.1e3))
        t0 = time.perf_counter()
        for s in range(W, W + K): step(s)
        for e in envs: e.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        say("stagger=%.2f own=%d hwq=%s " % (a.stagger, a.own, os.environ.get("GPU_MAX_HW_QUEUES", "-")) + "%-6s arenas %6d  sub-batches %d: %9.2f us per %d arena-steps   %.4g env-steps/s" % (name, A, k, el / K * 1e6, A, A * 4 * K / el))
        for e in envs: e.close()
        del envs, obs, dxdy, act
