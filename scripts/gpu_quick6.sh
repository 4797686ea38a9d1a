#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-q6}; mkdir -p $O; cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "lockstep or mode6" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for w in C3m6 C1 C2; do timeout 300 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-large > $O/bench_$w.json 2> $O/bench_$w.err; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
AGARCL_NO_FRONT=1 timeout 300 python scripts/gpu_phase6.py 2>&1 | tee $O/phase.txt | head -20
