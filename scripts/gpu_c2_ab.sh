#!/bin/bash
# The headline workload (C2) for several builds of the library on ONE box:  scripts/gpu_c2_ab.sh <variant> [<variant> ...]   (product | build_variants/lib_<NAME>.so)
for r in 1 2 3; do for v in "$@"; do if [ $v = product ]; then unset AGARCL_HIP_SO; else export AGARCL_HIP_SO=$PWD/build_variants/lib_$v.so; fi
  for a in 4096 65536; do python bench.py --arenas $a --steps 1000 --warmup 200 --no-cpu-baseline --no-large --no-full 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $a, 'arenas: %.2f us per step' % (b['ms_per_step']*1e3))"; done; done; done
