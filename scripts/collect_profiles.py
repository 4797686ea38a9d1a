"""Condenses scripts/profile_round.sh's rocprofv3 output into the files that get committed under profiles/.

    python3 scripts/collect_profiles.py <tag>       reads gpurun_out/prof_<tag>/, writes gpurun_out/profiles_<tag>/

<tag>_pmc_traffic.json: per run, HBM bytes per env step = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 summed over the step's kernels
(every launch of the run, warm-up included) / number of steps.  The x2 on fetches is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-byte requests at 64 bytes.  The file carries the
sha of the kernel source it was recorded on; bench.py quotes it only while that sha matches."""
import collections
import csv
import glob
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEP_KERNELS = ("k_fused", "k_quiet", "k_step", "k_order", "k_grid_zero", "k_grid_obs", "k_screen_obs", "k_ram_obs")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    os.makedirs(dst, exist_ok=True)
    import bench
    runs = {}
    for d in sorted(glob.glob(os.path.join(src, "*_*"))):
        if not os.path.isdir(d):
            continue
        name = os.path.basename(d)
        w, a = name.rsplit("_", 1)
        ent = {"workload": w, "arenas": int(a)}
        for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
            ks = {}
            for r in csv.DictReader(open(f)):
                for k in STEP_KERNELS:
                    if k in r["Name"]:
                        ks[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
            ent["kernel_stats"] = ks
        bj = os.path.join(src, name + ".bench.json")
        try:
            line = [l for l in open(bj).read().splitlines() if l.startswith("{")][-1]
            b = json.loads(line)
            json.dump(b, open(os.path.join(dst, "%s_%s_bench.json" % (tag, name)), "w"), indent=1)
            ent["steps_total"] = b["steps"] + b["warmup"]
            ent["bench_ms_per_step_under_profiler"] = b["ms_per_step"]
            ent["requested_bytes_per_step"] = b["roofline"]["requested_bytes_per_step"]
        except Exception as ex:
            ent["bench_error"] = str(ex)
        tot = collections.defaultdict(float); n = collections.defaultdict(int)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != c:
                        continue
                    for k in STEP_KERNELS:
                        if k in r["Kernel_Name"]:
                            tot[(k, c)] += float(r["Counter_Value"]); n[(k, c)] += 1
        sq = collections.defaultdict(float); sqn = collections.defaultdict(int)
        for f in glob.glob(os.path.join(d, "SQ", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                for k in STEP_KERNELS:
                    if k in r["Kernel_Name"]:
                        sq[(k, r["Counter_Name"])] += float(r["Counter_Value"]); sqn[(k, r["Counter_Name"])] += 1
        if sq:  # average per launch; derived: share of the VALU issue slots used and lanes busy per VALU instruction
            per = {}
            for (k, c), v in sq.items():
                per.setdefault(k, {})[c] = v / max(sqn[(k, c)], 1)
            for k, cnt in per.items():
                us = ent.get("kernel_stats", {}).get(k, {}).get("avg_us")
                if us and "SQ_INSTS_VALU" in cnt:
                    # 1024 SIMDs, one wave64 VALU instruction per 2 cycles each, 2.4 GHz nominal
                    cnt["valu_issue_share_at_2.4GHz"] = cnt["SQ_INSTS_VALU"] * 2.0 / (us * 1e-6 * 2.4e9 * 1024)
                if cnt.get("SQ_ACTIVE_INST_VALU"):
                    cnt["thread_cycles_per_active_valu_cycle"] = cnt.get("SQ_THREAD_CYCLES_VALU", 0.0) / cnt["SQ_ACTIVE_INST_VALU"]   # raw ratio (both in the counters' own units)
            ent["sq_per_launch"] = per
            # the dominant kernel's issue-rate figures, as bench.py quotes them (roofline.frac_valu_issue / mean_wave_residency / clock_ghz_measured):
            #   clock_ghz = SQ_BUSY_CYCLES / 32 / kernel time (the counter sums 32 shader engines' busy cycles);
            #   frac_valu_issue = SQ_INSTS_VALU x 2 cycles / (kernel time x clock x 1024 SIMDs): share of the VALU issue slots used;
            #   mean_wave_residency = (SQ_WAVE_CYCLES x 4 / wavefronts of the launch) / (SQ_BUSY_CYCLES / 32): mean time a wavefront is resident /
            #   the launch's duration (k_step only: one wavefront per arena, at most 4096 resident slots)
            ks = ent.get("kernel_stats", {})
            dom = max((k for k in per if k in ks), key=lambda k: ks[k]["total_ms"], default=None)
            if dom and per[dom].get("SQ_BUSY_CYCLES") and ks[dom].get("avg_us"):
                cnt, us = per[dom], ks[dom]["avg_us"]
                clock = cnt["SQ_BUSY_CYCLES"] / 32.0 / (us * 1e-6)
                iss = {"kernel": dom, "clock_ghz": clock / 1e9, "kernel_avg_us": us,
                       "frac_valu_issue": cnt.get("SQ_INSTS_VALU", 0.0) * 2.0 / (us * 1e-6 * clock * 1024.0)}
                if dom == "k_step" and cnt.get("SQ_WAVE_CYCLES"):
                    waves = float(min(int(a), 4096))
                    iss["mean_wave_residency"] = (cnt["SQ_WAVE_CYCLES"] * 4.0 / waves) / (cnt["SQ_BUSY_CYCLES"] / 32.0)
                ent["issue"] = iss
        if tot and ent.get("steps_total"):
            st = ent["steps_total"]
            fetch = sum(v for (k, c), v in tot.items() if c == "FETCH_SIZE") * 1024.0 / st
            write = sum(v for (k, c), v in tot.items() if c == "WRITE_SIZE") * 1024.0 / st
            ent["fetch_bytes_per_step_raw"] = fetch; ent["write_bytes_per_step"] = write
            ent["traffic_bytes_per_step"] = 2.0 * fetch + write
            ent["traffic_bytes_per_step_uncorrected"] = fetch + write
            ent["per_kernel_KB_per_launch"] = {"%s.%s" % k: tot[k] / max(n[k], 1) for k in tot}
            ent["launches"] = {"%s.%s" % k: n[k] for k in n}
        runs["%s@%s" % (w, a)] = ent
    out = {"source_sha": bench.source_sha(), "recorded": time.strftime("%Y-%m-%d") + " " + tag,
           "collection": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python3 bench.py --workload W --arenas A ...` "
                         "(scripts/profile_round.sh); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the step kernels / steps",
           "runs": runs}
    json.dump(out, open(os.path.join(dst, "%s_pmc_traffic.json" % tag), "w"), indent=1)
    for k, e in runs.items():
        print(k, {x: e.get(x) for x in ("traffic_bytes_per_step", "requested_bytes_per_step", "kernel_stats")})


if __name__ == "__main__":
    main()
