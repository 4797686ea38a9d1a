#!/bin/bash
# A/B of the compiler's scheduling strategy (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause) against the default build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for r in 1 2; do for v in base v1 v2; do
  so=$ROOT/agarcl_amd/libagarcl_hip.so; [ $v != base ] && so=$ROOT/agarcl_amd/libagarcl_hip_$v.so
  for w in "C2 --steps 1000 --warmup 100" "C3m6 --steps 150 --warmup 40" "C1 --steps 150 --warmup 40" "C2 --arenas 65536 --steps 300 --warmup 50"; do
    set -- $w; wl=$1; shift
    t=$(AGARCL_HIP_SO=$so python bench.py --workload $wl "$@" --no-cpu-baseline --no-large 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']*1e3))")
    echo "$v round $r: $wl $* -> $t us/step"
  done
done; done
