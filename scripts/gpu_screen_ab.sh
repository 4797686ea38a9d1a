#!/bin/bash
# k_screen_obs alone (scripts/gpu_screen_time.py) for several builds of the library on ONE box:  scripts/gpu_screen_ab.sh <variant> ...   (product | build_variants/lib_<NAME>.so)
for r in 1 2; do for v in "$@"; do if [ $v = product ]; then unset AGARCL_HIP_SO; else export AGARCL_HIP_SO=$PWD/build_variants/lib_$v.so; fi
  echo "== $v"; python scripts/gpu_screen_time.py 2>&1 | grep -v amdgpu.ids | tail -12 | grep -E "128x128x4|84x 84x4|task3.*84x 84x3"; done; done
