#!/bin/bash
# Parallel GPU soak: <procs> copies of scripts/gpu_soak.py with seeds <seed0>.. and <trials> trials each (the oracle side is host-bound, so
# several processes share the GPU).  A process that dies (e.g. a GPU fault) leaves its log under gpurun_out/soak_<seed>.log: keep it.
#   bash scripts/gpu_soak_par.sh <seed0> <procs> <trials> [ENV=VAL ...]
seed0=$1; procs=$2; trials=$3; shift 3
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
pids=()
for i in $(seq 0 $((procs - 1))); do
  s=$((seed0 + i))
  ( SOAK_VERBOSE=1 timeout 3000 python scripts/gpu_soak.py $s $trials > gpurun_out/soak_$s.log 2>&1; echo "exit $?" >> gpurun_out/soak_$s.log ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for i in $(seq 0 $((procs - 1))); do s=$((seed0 + i)); echo "== seed $s: $(grep -E 'soak done|MISMATCH|exit' gpurun_out/soak_$s.log | tr '\n' ' ')"; grep -i -E "aperture|fault|error" gpurun_out/soak_$s.log | head -3; done
