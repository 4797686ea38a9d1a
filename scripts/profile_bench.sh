#!/bin/bash
# rocprofv3 kernel-trace + stats of the headline bench (run on the GPU box through gpurun).
# Usage: scripts/profile_bench.sh <tag>   -> gpurun_out/prof_<tag>/ (copy the *_stats.csv into profiles/)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline > $OUT/bench_stdout.txt 2>&1
find $OUT -name '*kernel_stats.csv' -exec head -20 {} \;
tail -1 $OUT/bench_stdout.txt
