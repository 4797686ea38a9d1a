#!/bin/bash
# Second sweep of the front kernel's lanes per arena: finer arena counts, QG = 1 included, fused launch beside it.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abq2}; mkdir -p $O; cd $ROOT
for a in 6144 8192 12288 16384 32768 65536 131072 262144 524288; do
  [ $a -le 32768 ] && AGARCL_FUSED=1 timeout 300 python bench.py --arenas $a --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_${a}_fused.json 2> $O/err.txt
  for q in 8 4 2 1; do
    AGARCL_QUIET_QG=$q AGARCL_FUSED=0 timeout 300 python bench.py --arenas $a --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_${a}_qg$q.json 2> $O/err.txt
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/b_*.json"), key=lambda f: (int(os.path.basename(f).split("_")[1]), f)):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  kernel %.2f us" % (b["value"], b["ms_per_step"]*1e3, b["roofline"]["kernel_ms"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
