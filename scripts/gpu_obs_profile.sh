#!/bin/bash
# rocprofv3 kernel stats of the grid-observation kernels (full 8-channel config), through gpurun
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/obsprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/scripts/gpu_obs_timing.py full > $OUT/stdout.txt 2>&1
grep -E "k_grid" $OUT/p_kernel_stats.csv | cut -c1-160
