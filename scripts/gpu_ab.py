"""One A/B tool for the HIP library (replaces the scripts/gpu_ab_* family): interleaved runs of bench.py over
    variants  x  workloads  x  repetitions
where a variant is a build of the library (build_variants/lib_<NAME>.so, or `product`) optionally with launch pins, and a workload is
<bench workload>[@arenas].  Prints the HIP-event time per step of every run and the per-cell median.

    python scripts/gpu_ab.py --variants product,MATRIX,DIRTY --workloads C3m6@4096,C3m6@32768,mid@4096,C2@4096,C2@65536 --reps 3
    python scripts/gpu_ab.py --variants product,product:AGARCL_FUSED=0 --workloads C2@16384 --steps 1000
Every bench run is a child process (bench.py's own contract: untimed warm-up, barrier + synchronise, HIP events on the launch stream)."""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="product"); ap.add_argument("--workloads", default="C2@4096")
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--steps", type=int, default=0); ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--out", default="ab")
a = ap.parse_args()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", a.out + ".log"), "a")
def say(s):
    print(s, flush=True); log.write(s + "\n"); log.flush()
cells = {}
for rep in range(a.reps):
    for wl in a.workloads.split(","):
        name, arenas = (wl.split("@") + ["4096"])[:2]
        heavy = name in ("C3m6", "C5", "C5s", "C1") or int(arenas) > 16384
        steps = a.steps or (100 if heavy else 1000); warm = a.warmup or (400 if name == "mid" else (20 if heavy else 100))
        for var in a.variants.split(","):
            lib, *pins = var.split(":")
            env = dict(os.environ)
            if lib != "product": env["AGARCL_HIP_SO"] = os.path.join(ROOT, "build_variants", "lib_%s.so" % lib)
            for kv in pins: k, v = kv.split("="); env[k] = v
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", name, "--arenas", arenas, "--steps", str(steps), "--warmup", str(warm),
                   "--no-cpu-baseline", "--no-large", "--no-full"]
            try:
                p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
                b = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
                us = b["roofline"]["kernel_ms"] * 1e3
                cells.setdefault((wl, var), []).append(us)
                say("%-14s %-28s rep %d: %9.2f us/step (HIP events)  %9.2f us wall  %.4g env-steps/s  flags %d" % (wl, var, rep, us, b["ms_per_step"] * 1e3, b["value"], b["capacity_flags_raised"]))
            except Exception as ex:
                say("%-14s %-28s rep %d: FAILED %s | %s" % (wl, var, rep, ex, (p.stderr.strip().splitlines() or [""])[-1][:200] if 'p' in dir() else ""))
say("---- medians (us per step, HIP events)")
for (wl, var), v in cells.items():
    say("%-14s %-28s %9.2f   (min %.2f, n %d)" % (wl, var, statistics.median(v), min(v), len(v)))
