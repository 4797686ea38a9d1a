#!/bin/bash
# interleaved A/B of two builds of the library on the C2 workload: scripts/gpu_ab_so.sh <tag> <other.so> [arenas...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abso}; OTHER=$ROOT/$2; shift; shift; mkdir -p $O; cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "quiet or front or 4096 or C2 or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
for rep in 1 2 3; do for a in ${@:-4096 65536}; do
  for v in new other; do
    so=$ROOT/agarcl_amd/libagarcl_hip.so; [ $v = other ] && so=$OTHER
    AGARCL_HIP_SO=$so timeout 300 python bench.py --arenas $a --steps 1000 --warmup 100 --no-cpu-baseline --no-large > $O/b_${a}_${v}_$rep.json 2> $O/b_${a}_${v}_$rep.err
    AGARCL_HIP_SO=$so timeout 300 python bench.py --arenas $a --steps 20 --warmup 5 --no-cpu-baseline --no-large > $O/d_${a}_${v}_$rep.json 2> /dev/null
  done
done; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/[bd]_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  kernel %.2f us" % (b["value"], b["ms_per_step"]*1e3, b["roofline"]["kernel_ms"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
