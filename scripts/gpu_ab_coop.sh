#!/bin/bash
# A/B on one box: cooperative sweeps on / off, mode 6 and C1, plus the phase profile (front kernel pinned off).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-ab}; mkdir -p $O; cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "lockstep or mode6" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for coop in 1 0; do for w in C3m6 C1; do
  AGARCL_COOP=$coop timeout 300 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-large > $O/bench_${w}_coop$coop.json 2> $O/bench_${w}_coop$coop.err
done; done
AGARCL_COOP=1 timeout 300 python bench.py --workload C3m6 --arenas 32768 --steps 60 --warmup 20 --no-cpu-baseline --no-large > $O/bench_C3m6_32768_coop1.json 2>/dev/null
AGARCL_COOP=0 timeout 300 python bench.py --workload C3m6 --arenas 32768 --steps 60 --warmup 20 --no-cpu-baseline --no-large > $O/bench_C3m6_32768_coop0.json 2>/dev/null
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
for coop in 1 0; do echo "== phase profile coop=$coop"; AGARCL_COOP=$coop AGARCL_NO_FRONT=1 timeout 300 python scripts/gpu_phase6.py 2>&1 | tee $O/phase_coop$coop.txt; done
