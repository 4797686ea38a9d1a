#!/bin/bash
# One GPU session of the round: parity tests, headline + driver-style + side-workload bench lines.  scripts/gpu_round.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver20.json 2>> $O/bench_default.err
for w in C3m0 C3m6 mid C5 C5s C1; do python bench.py --workload $w --steps 200 --warmup $([ $w = mid ] && echo 400 || echo 40) --no-cpu-baseline --no-full > $O/bench_$w.json 2> $O/bench_$w.err; done
for a in 1024 16384 65536 262144; do python bench.py --arenas $a --steps 200 --warmup 40 --no-cpu-baseline --no-large > $O/bench_C2_$a.json 2> $O/bench_C2_$a.err; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        r=b["roofline"]
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  frac %.3f  model_speedup %.3f  req %.3g B  work %s" % (b["value"], b["ms_per_step"]*1e3, r["frac"], r["model_speedup"], r["requested_bytes_per_step"], r["work_per_step"]))
        if "roofline_large" in b: print("   large:", {k:b["roofline_large"].get(k) for k in ("value_env_steps_per_s","ms_per_step","frac","achieved","requested_bytes_per_step")})
        if "cpu_baseline" in b: print("   cpu:", b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"], b["cpu_baseline"].get("c1_ticks_per_s_1core"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:] if os.path.exists(f.replace(".json",".err")) else "")
PY
