#!/bin/bash
# Short GPU check of a build: the GPU suite, then the step workloads of the round's targets.  scripts/gpu_check.sh <tag> [pytest -k expr]
TAG=${1:-chk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$2" > $O/pytest.log 2>&1; else timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; fi
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for w in C2 C3m6 mid C1; do
  timeout 300 python bench.py --workload $w --steps 200 --warmup $([ $w = mid ] && echo 400 || echo 40) --no-cpu-baseline --no-full --no-large > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
try:
    b=json.loads([l for l in open("$O/bench_$w.json").read().splitlines() if l.startswith("{")][-1])
    print("$w", "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3))
except Exception as e: print("$w ERR", e, open("$O/bench_$w.err").read()[-600:])
PY
done
