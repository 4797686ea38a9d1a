#!/bin/bash
# Lanes per arena of the single-launch step at larger arena counts (which wavefront count to aim for: 1024 or 2048).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for a in 12288 16384 32768 65536 131072; do for q in 16 8 4 2 1; do
  w=$(( a * q / 64 )); [ $w -gt 4096 ] && continue; [ $w -lt 512 ] && continue
  r=$(AGARCL_FUSED=1 AGARCL_FUSED_QG=$q python bench.py --arenas $a --steps 400 --warmup 100 --no-cpu-baseline --no-large 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']*1e3))")
  echo "arenas $a lanes/arena $q ($w wavefronts): $r us/step"
done; done
