#!/bin/bash
# Copies one GPU session's results into profiles/ under the round's names:  scripts/adopt_profiles.sh <session tag> [round tag]
#   gpurun_out/profiles_<tag>/<tag>_*  (scripts/profile_round.sh + collect_profiles.py)  -> profiles/<round>_*
#   gpurun_out/<tag>/bench_default.json, bench_driver20.json (scripts/gpu_round.sh)       -> profiles/<round>_bench_*_unprofiled.json
# and, with --resources, rebuilds the kernel resource table (hipcc -Rpass-analysis=kernel-resource-usage, ~2.5 min).
set -e
TAG=$1; ROUND=${2:-r03}; ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
for f in gpurun_out/profiles_$TAG/${TAG}_*; do cp $f profiles/${ROUND}_${f#gpurun_out/profiles_$TAG/${TAG}_}; done
for n in default driver20; do
  [ -f gpurun_out/$TAG/bench_$n.json ] && grep '^{' gpurun_out/$TAG/bench_$n.json | tail -1 > profiles/${ROUND}_bench_${n}_unprofiled.json
done
if [ "${3:-}" = "--resources" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -Rpass-analysis=kernel-resource-usage -o /tmp/agarcl_res.so agarcl_amd/csrc/agar_engine.hip 2> /tmp/agarcl_res.txt
  python3 - <<PY
import re
t = open("/tmp/agarcl_res.txt").read()
out = ["# hipcc -Rpass-analysis=kernel-resource-usage of agarcl_amd/csrc/agar_engine.hip (flags of agarcl_amd/build.py), source sha " + __import__("subprocess").check_output(["python3", "-c", "import bench; print(bench.source_sha())"]).decode().strip(),
       "# kernel | VGPRs | AGPRs | SGPRs | scratch bytes/lane | occupancy waves/SIMD | LDS bytes/block"]
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    out.append(" | ".join([b.split()[0], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")]))
open("profiles/${ROUND}_kernel_resource_usage.txt", "w").write("\n".join(out) + "\n")
print(len(out) - 2, "kernels")
PY
fi
python3 -c "import json,bench; t=json.load(open('profiles/${ROUND}_pmc_traffic.json')); print('pmc sha', t['source_sha'], 'current', bench.source_sha())"
