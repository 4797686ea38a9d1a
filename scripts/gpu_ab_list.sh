#!/bin/bash
# A/B: fused single launch vs k_quiet + work-list k_step (k_quiet at 138 VGPRs or capped at 128), several arena counts; mode 6 too
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abl}; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "lockstep or mode6 or quiet or front or adapts or 4096" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for a in 4096 16384 65536 262144; do
  for v in fused twok twok_q4; do
    so=$ROOT/agarcl_amd/libagarcl_hip.so; fu=0
    [ $v = fused ] && fu=1
    [ $v = twok_q4 ] && so=$ROOT/agarcl_amd/libagarcl_hip_q4.so
    AGARCL_HIP_SO=$so AGARCL_FUSED=$fu timeout 300 python bench.py --arenas $a --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_${a}_$v.json 2> $O/b_${a}_$v.err
  done
done
for a in 4096 32768; do timeout 300 python bench.py --workload C3m6 --arenas $a --steps 100 --warmup 30 --no-cpu-baseline --no-large > $O/m6_$a.json 2> $O/m6_$a.err; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
