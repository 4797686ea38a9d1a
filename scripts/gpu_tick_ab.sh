#!/bin/bash
# The several-player rows of the bench for several builds of the library on ONE box:  scripts/gpu_tick_ab.sh <variant> ...   (product | build_variants/lib_<NAME>.so)
for r in 1 2; do for v in "$@"; do if [ $v = product ]; then unset AGARCL_HIP_SO; else export AGARCL_HIP_SO=$PWD/build_variants/lib_$v.so; fi
  for w in tick10 tick30 C1; do python bench.py --workload $w --steps 150 --warmup 40 --no-cpu-baseline --no-large --no-full 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w %.2f us per step' % (b['ms_per_step']*1e3))"; done; done; done
