"""Variants of the closing synchronisation of a 20-step timed region (see gpu_driver20.py)."""
import sys, time
sys.path.insert(0, '.')
import torch
from agarcl_amd.vec_env import VecEnvironment
import bench

A = 4096
cfg = dict(bench.CFG)
env = VecEnvironment(A, strict_flags=False, **cfg)
eng = env.engine
g = torch.Generator(device=env.device); g.manual_seed(1234)
dx = (torch.rand((25, A, 1, 2), generator=g, device=env.device) * 2 - 1).contiguous()
ac = torch.zeros((25, A, 1), dtype=torch.int32, device=env.device)
st = torch.cuda.current_stream()
for variant in ("events+device_sync", "no_events", "events+stream_sync", "no_events+stream_sync", "event_sync", "events+device_sync"):
    tot = []
    for rep in range(5):
        env.seed(base_seed=10000); env.reset(reset_ids=True)
        for k in range(5):
            eng.set_actions_device(dx[k].data_ptr(), ac[k].data_ptr()); eng.step(4)
        torch.cuda.synchronize(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if "no_events" not in variant: e0.record()
        for k in range(5, 25):
            eng.set_actions_device(dx[k].data_ptr(), ac[k].data_ptr()); eng.step(4)
        if "no_events" not in variant: e1.record()
        t1 = time.perf_counter()
        if variant == "event_sync": e1.synchronize()
        if "stream_sync" in variant: st.synchronize()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tot.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6 / 20))
    print("%-24s" % variant, " | ".join("enq %.0f first %.0f rest %.0f => %.2f us/step" % t for t in tot[1:]))
env.close()
