"""Prints the measurement tables of DESIGN.md sections 4.4 / 5 from the round's committed records (profiles/r05_*): the bench line at the
driver's arguments and at its defaults, the PMC / SQ passes, the kernel resource table.  python scripts/design_tables.py [round tag]
(DESIGN.md quotes these figures; regenerate them with this script when profiles/ is re-recorded instead of editing numbers by hand.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
drv = json.load(open(P("bench_driver20_unprofiled.json"))); dft = json.load(open(P("bench_default_unprofiled.json"))); pmc = json.load(open(P("pmc_traffic.json")))
cols = drv["roofline"]["by_workload_columns"]
print("kernel source sha %s, recorded %s\n" % (pmc["source_sha"], pmc["recorded"]))
print("| workload | us per step (driver window: 20-150 steps) | env-steps/s | us per step (default run) | PMC traffic per step | traffic / requested | frac of 8 TB/s | VALU issue share | mean wave residency | clock GHz | reference CPU env-steps/s (cores) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
bd, bf = drv["roofline"]["by_workload"], dft["roofline"]["by_workload"]
def g(row, name): return row[cols.index(name)] if isinstance(row, list) else None
for k in bd:
    r, r2 = bd[k], bf.get(k)
    run = pmc["runs"].get(k.split("/")[0]) if "/" not in k else None
    iss = (run or {}).get("issue") or {}
    tr = (run or {}).get("traffic_bytes_per_step"); rq = (run or {}).get("requested_bytes_per_step")
    ms = g(r, "ms_per_step")
    print("| %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        k, "%.1f" % (ms * 1e3) if ms else "-", "%.4g" % g(r, "env_steps_per_s") if ms else "-", "%.1f" % (g(r2, "ms_per_step") * 1e3) if isinstance(r2, list) else "-",
        "%.1f MB" % (tr / 1e6) if tr else "-", ("%.2f" % g(r, "traffic_over_requested")) if g(r, "traffic_over_requested") else ("%.2f" % (tr / rq) if tr and rq else "-"), "%.3f" % (tr / (ms * 1e-3) / 8e12) if tr and ms else "-",
        "%.2f" % iss["frac_valu_issue"] if iss.get("frac_valu_issue") else "-", "%.2f" % iss["mean_wave_residency"] if iss.get("mean_wave_residency") else "-",
        "%.2f" % iss["clock_ghz"] if iss.get("clock_ghz") else "-",
        ("%.3g (%d)" % (g(r, "cpu_reference_env_steps_per_s"), g(r, "cpu_reference_cores"))) if g(r, "cpu_reference_env_steps_per_s") else "-"))
print("\nheadline: driver window %.4g env-steps/s (%.2f us per step), default run (1000 steps) %.4g (%.2f us)" % (drv["value"], drv["ms_per_step"] * 1e3, dft["value"], dft["ms_per_step"] * 1e3))
print("vector surface (host us per step, total us per step):", {k: (round(v["host_us_per_step"], 1), round(v["us_per_step"], 1)) for k, v in drv["roofline"]["gym_vector"].items()})
print("\nkernel avg under rocprofv3 (us):")
for k, e in pmc["runs"].items():
    print("  %-12s %s" % (k, {kk: round(v["avg_us"], 1) for kk, v in e.get("kernel_stats", {}).items()}))
