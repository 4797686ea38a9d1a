#!/bin/bash
# Quick bench lines only (no tests): headline, driver-style, side workloads, arena counts.  scripts/gpu_quick_bench.sh <tag>
TAG=${1:-q}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O; cd $ROOT
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-large > $O/bench_C2_4096.json 2> $O/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large > $O/bench_driver20.json 2>> $O/err.txt
for w in C3m0 C3m6 C1; do python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-large > $O/bench_$w.json 2>> $O/err.txt; done
for a in 1024 8192 12288 16384 32768 65536 131072 262144; do python bench.py --arenas $a --steps 200 --warmup 40 --no-cpu-baseline --no-large > $O/bench_C2_$a.json 2>> $O/err.txt; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  kernel %.2f us" % (b["value"], b["ms_per_step"]*1e3, b["roofline"]["kernel_ms"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
