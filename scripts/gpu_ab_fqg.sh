#!/bin/bash
# Lanes per arena of the single-launch step (AGARCL_FUSED_QG) at small arena counts: steady state and driver-style timing.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for a in 1024 2048 4096 8192; do for q in 8 16 32 64; do
  r=$(AGARCL_FUSED_QG=$q python bench.py --arenas $a --steps 1000 --warmup 100 --no-cpu-baseline --no-large 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']*1e3))")
  d1=$(AGARCL_FUSED_QG=$q python bench.py --arenas $a --steps 20 --warmup 5 --no-cpu-baseline --no-large 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f (kernel %.2f)' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))")
  echo "arenas $a lanes/arena $q: steady $r us/step, driver-style $d1"
done; done
