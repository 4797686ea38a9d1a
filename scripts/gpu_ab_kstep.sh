#!/bin/bash
# A/B of k_step's register cap (waves per SIMD) on the full ruleset and the C1 population
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abk}; mkdir -p $O; cd $ROOT
for rep in 1 2; do for v in base w5 w6 w8; do
  so=$ROOT/agarcl_amd/libagarcl_hip.so; [ $v != base ] && so=$ROOT/agarcl_amd/libagarcl_hip_$v.so
  for w in C3m6 C1; do AGARCL_HIP_SO=$so timeout 300 python bench.py --workload $w --steps 150 --warmup 40 --no-cpu-baseline --no-large > $O/${w}_${v}_$rep.json 2> $O/${w}_${v}_$rep.err; done
  AGARCL_HIP_SO=$so timeout 300 python bench.py --workload C3m6 --arenas 32768 --steps 40 --warmup 10 --no-cpu-baseline --no-large > $O/C3m6x32768_${v}_$rep.json 2>/dev/null
done; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
