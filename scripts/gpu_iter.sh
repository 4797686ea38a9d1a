#!/bin/bash
# quick perf iteration on the GPU box: a parity subset, then short bench lines.  scripts/gpu_iter.sh <tag> [pytest -k expr]
TAG=${1:-it}; KEXPR=${2:-"lockstep or golden or mode6 or c1_population"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for w in C3m6 C1 C2; do timeout 300 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-large > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python bench.py --workload C3m6 --arenas 32768 --steps 100 --warmup 40 --no-cpu-baseline --no-large > $O/bench_C3m6_32768.json 2> $O/bench_C3m6_32768.err
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); r=b["roofline"]
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  frac %.3f" % (b["value"], b["ms_per_step"]*1e3, r["frac"]))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
