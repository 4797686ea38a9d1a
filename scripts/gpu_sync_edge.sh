#!/bin/bash
# The driver's 20-step window (bench.py --steps 20 --warmup 5) under different host wait modes of the HIP runtime: how much of its ~35 us of edges is the
# wake-up of a blocked host thread.  scripts/gpu_sync_edge.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-full 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f us/step  %.3f G' % (d['ms_per_step']*1e3, d['value']/1e9))"; }
for rep in 1 2 3; do
  echo "default:                         $(one)"
  echo "ROC_ACTIVE_WAIT_TIMEOUT=1000:    $(ROC_ACTIVE_WAIT_TIMEOUT=1000 one)"
  echo "HIP spin flag (AGARCL_BENCH_SPIN=1): $(AGARCL_BENCH_SPIN=1 one)"
done
