#!/bin/bash
# The round's measurement set (run through gpurun): headline bench + rocprofv3 kernel stats + the other SURVEY 8(d)
# workloads + batch-size sweep + PMC traffic.  Everything lands in gpurun_out/<tag>_*; copy into profiles/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 300 python bench.py --steps 1000 --warmup 100 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
: > gpurun_out/${TAG}_workloads.jsonl
for w in C3m0 C3m6 C5 C5s; do timeout 300 python bench.py --steps 200 --warmup 250 --workload $w 2>/dev/null | tail -1 >> gpurun_out/${TAG}_workloads.jsonl; done
for a in 1024 16384 65536; do timeout 300 python bench.py --steps 200 --warmup 40 --arenas $a --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_workloads.jsonl; done
timeout 400 bash scripts/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
cp gpurun_out/prof_$TAG/bench_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
timeout 500 bash scripts/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
cut -c1-400 gpurun_out/${TAG}_bench.json; cut -c1-260 gpurun_out/${TAG}_workloads.jsonl; head -4 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200; cat gpurun_out/pmc_traffic_$TAG.json
