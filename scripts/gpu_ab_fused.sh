#!/bin/bash
# A/B of k_fused's register cap (waves per SIMD 2 / 3 / 4) and of the two-kernel step at several arena counts (one box, interleaved)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abf}; mkdir -p $O; cd $ROOT
for a in 4096 16384 65536 262144; do
  for v in base f3 f4 twok; do
    so=$ROOT/agarcl_amd/libagarcl_hip.so; fu=""
    [ $v = f3 ] && so=$ROOT/agarcl_amd/libagarcl_hip_f3.so
    [ $v = f4 ] && so=$ROOT/agarcl_amd/libagarcl_hip_f4.so
    [ $v = twok ] && fu=0
    AGARCL_HIP_SO=$so AGARCL_FUSED=$fu timeout 300 python bench.py --arenas $a --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_${a}_$v.json 2> $O/b_${a}_$v.err
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/b_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
