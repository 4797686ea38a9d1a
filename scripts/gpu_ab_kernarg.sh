#!/bin/bash
# A/B on one box: descriptor pointers as (preloaded) kernel arguments.  base = libagarcl_hip.so; v1 = preload flag only;
# v2 = hot pointers as arguments; v3 = both.  Interleaved, two rounds.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abk}; mkdir -p $O; cd $ROOT
for r in 1 2; do for v in base v1 v2 v3; do
  so=$ROOT/agarcl_amd/libagarcl_hip.so; [ $v != base ] && so=$ROOT/agarcl_amd/libagarcl_hip_$v.so
  AGARCL_HIP_SO=$so timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-large > $O/b_4096_${v}_$r.json 2> $O/err.txt
  AGARCL_HIP_SO=$so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large > $O/b_drv20_${v}_$r.json 2> $O/err.txt
  AGARCL_HIP_SO=$so timeout 300 python bench.py --arenas 65536 --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_65536_${v}_$r.json 2> $O/err.txt
done; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/b_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  kernel %.2f us" % (b["value"], b["ms_per_step"]*1e3, b["roofline"]["kernel_ms"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
