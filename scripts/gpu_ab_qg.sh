#!/bin/bash
# Lanes per arena of the lean front kernel (AGARCL_QUIET_QG = 16 / 8 / 4 / 2) in the two-kernel step, C2, several arena counts.
# Fewer lanes per arena = fewer wavefronts for the same arenas (the quiet tick is per-lane code; a pellet pass is wave-wide anyway).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abq}; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or lockstep or golden" > $O/pytest16.log 2>&1; echo "pytest rc=$?" >> $O/pytest16.log; tail -2 $O/pytest16.log
for q in 4 2; do AGARCL_QUIET_QG=$q AGARCL_FUSED=0 timeout 900 python -m pytest tests -m gpu -x -q -k "parity or lockstep or golden" > $O/pytest$q.log 2>&1; echo "pytest qg=$q rc=$?" >> $O/pytest$q.log; tail -2 $O/pytest$q.log; done
for a in 4096 16384 65536 262144; do
  timeout 300 python bench.py --arenas $a --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_${a}_default.json 2> $O/err.txt
  for q in 16 8 4 2; do
    AGARCL_QUIET_QG=$q AGARCL_FUSED=0 timeout 300 python bench.py --arenas $a --steps 300 --warmup 50 --no-cpu-baseline --no-large > $O/b_${a}_qg$q.json 2> $O/err.txt
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/b_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  kernel %.2f us" % (b["value"], b["ms_per_step"]*1e3, b["roofline"]["kernel_ms"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
