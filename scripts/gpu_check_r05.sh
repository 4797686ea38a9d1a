#!/bin/bash
# A kernel change between two full sessions: the GPU suite, the multi-player rows of the bench, the phase split.  scripts/gpu_check_r05.sh <tag>
TAG=${1:-chk}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O; cd $ROOT
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for w in tick5 tick10 tick20 tick30 C1 C3m6; do python bench.py --workload $w --steps 150 --warmup 40 --no-cpu-baseline --no-large --no-full > $O/bench_$w.json 2>> $O/err.txt; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g %s  %.2f us/step" % (b["value"], b["unit"], b["ms_per_step"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
[ -f build_variants/lib_PROF.so ] && python scripts/gpu_phase_multi.py > $O/phase_multi.log 2>&1; cat $O/phase_multi.log
