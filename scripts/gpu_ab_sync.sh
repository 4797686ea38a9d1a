#!/bin/bash
# how the host waits for the GPU (interrupt vs polling signals) in the driver-style short run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-abs}; mkdir -p $O; cd $ROOT
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large > $O/d_int_$rep.json 2>/dev/null
  HSA_ENABLE_INTERRUPT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large > $O/d_poll_$rep.json 2>/dev/null
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-large > $O/l_int_$rep.json 2>/dev/null
  HSA_ENABLE_INTERRUPT=0 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-large > $O/l_poll_$rep.json 2>/dev/null
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "%.4g env-steps/s  %.2f us/step  kernel %.2f us" % (b["value"], b["ms_per_step"]*1e3, b["roofline"]["kernel_ms"]*1e3))
    except Exception as e: print(f, "ERR", e)
PY
