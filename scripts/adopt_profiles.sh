#!/bin/bash
# Copies one GPU session's results into profiles/ under the round's names:  scripts/adopt_profiles.sh <session tag> [round tag]
#   gpurun_out/profiles_<tag>/<tag>_*  (scripts/profile_round.sh + collect_profiles.py)  -> profiles/<round>_*
#   gpurun_out/<tag>/bench_default.json, bench_driver20.json (scripts/gpu_session.sh)       -> profiles/<round>_bench_*_unprofiled.json
# and, with --resources, rebuilds the kernel resource table (hipcc -Rpass-analysis=kernel-resource-usage, ~2.5 min).
set -e
TAG=$1; ROUND=${2:-r06}; ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
for f in gpurun_out/profiles_$TAG/${TAG}_*; do cp $f profiles/${ROUND}_${f#gpurun_out/profiles_$TAG/${TAG}_}; done
for n in default driver20; do
  [ -f gpurun_out/$TAG/bench_$n.json ] && grep '^{' gpurun_out/$TAG/bench_$n.json | tail -1 > profiles/${ROUND}_bench_${n}_unprofiled.json
  [ -f gpurun_out/$TAG/bench_${n}_full.json ] && cp gpurun_out/$TAG/bench_${n}_full.json profiles/${ROUND}_bench_${n}_full.json
done
if [ "${3:-}" = "--resources" ]; then   # the kernel resource table, unit by unit with the build's own flags (~40 s)
  python3 -m agarcl_amd.build --resources profiles/${ROUND}_kernel_resource_usage.txt
fi
python3 -c "import json,bench; t=json.load(open('profiles/${ROUND}_pmc_traffic.json')); print('pmc sha', t['source_sha'], 'current', bench.source_sha())"
