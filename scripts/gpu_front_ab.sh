#!/bin/bash
# The two-kernel step with and without its front launch (AGARCL_NO_FRONT=1), us per step:  scripts/gpu_front_ab.sh [workload:arenas ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for wa in ${@:-task1:4096 task2:4096 mid:4096 task3:4096 mid:16384 task1:16384}; do
  IFS=: read W A <<< "$wa"; WARM=60; [ "$W" = "mid" ] && WARM=400
  for nf in 0 1; do
    ms=$(AGARCL_NO_FRONT=$nf python bench.py --workload $W --arenas $A --steps 150 --warmup $WARM --no-cpu-baseline --no-large --no-full 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "$W@$A no_front=$nf  $ms ms/step"
  done
done
