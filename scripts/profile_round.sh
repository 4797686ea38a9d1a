#!/bin/bash
# Records the round's profiles on the GPU box (run through gpurun):  scripts/profile_round.sh <tag> [runs...]
# For every run "WORKLOAD:ARENAS:STEPS" three rocprofv3 passes over the SAME command (python3 bench.py ..., the program
# directly after `--`): --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE on their own (they do not fit one
# pass on gfx950; counters only, no tracing domains).  scripts/collect_profiles.py turns the output into
# gpurun_out/profiles_<tag>/ : <tag>_<run>_kernel_stats.csv, <tag>_<run>_bench.json and <tag>_pmc_traffic.json -- copy
# those into profiles/ and commit them.
set -u
TAG=${1:-r06}; shift
RUNS=${@:-"C2:4096:400 C2:65536:200 C2:262144:100 C3m6:4096:200 C3m6:32768:60 mid:4096:200 C5:4096:100 C5s:4096:100 C1:4096:200 C1r:4096:200 task3:4096:100 task6:4096:60"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for run in $RUNS; do
  IFS=: read W A S <<< "$run"
  WARM=40; [ "$W" = "mid" ] && WARM=400
  ARGS="--workload $W --arenas $A --steps $S --warmup $WARM --no-cpu-baseline --no-large --no-full"
  name=${W}_${A}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/kt -o p -- python3 $ROOT/bench.py $ARGS > $OUT/$name.bench.json 2> $OUT/$name.kt.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/$name/$c -o p -- python3 $ROOT/bench.py $ARGS > $OUT/$name.$c.json 2> $OUT/$name.$c.err
  done
  if [ "$A" = "4096" ] || [ "$W:$A" = "C3m6:32768" ] || [ "$W:$A" = "C2:65536" ]; then   # what the shader engines were doing (8 SQ counters = one pass): instruction mix, lane utilisation
    timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/$name/SQ -o p -- python3 $ROOT/bench.py $ARGS > $OUT/$name.SQ.json 2> $OUT/$name.SQ.err
  fi
done
python3 $ROOT/scripts/collect_profiles.py $TAG && rm -rf $OUT   # (the raw rocprofv3 output is scratch: gpurun_out/ copies back at most 64 MiB)
