#!/bin/bash
# kernel-level timing of k_quiet / k_step for a few C2 variants (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "1.0 4 4096" "0.0 4 4096" "1.0 1 4096" "1.0 16 4096" "1.0 4 1024" "1.0 4 16384"; do
  OUT=$ROOT/gpurun_out/qp; rm -rf $OUT; mkdir -p $OUT
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/scripts/gpu_quiet_probe.py $cfg > $OUT/stdout.txt 2>&1
  echo "== move ticks arenas = $cfg"
  grep -E "k_quiet|k_step" $OUT/p_kernel_stats.csv | cut -d, -f1-3,7- | sed -E 's/\(AgState[^"]*"/"/'
done
