#!/bin/bash
# HBM traffic of one env step (k_fused, or k_quiet + k_step when AGARCL_FUSED=0) from the TCC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one pass on gfx950), counters only (no tracing domains).  Workload: C2, 4096 arenas, 4 ticks per
# step, a fresh random direction every step (scripts/gpu_quiet_probe.py 1.0 4 4096 random = 300 steps).
# Output: gpurun_out/pmc_traffic_<tag>.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$ROOT/gpurun_out/traffic_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- python3 $ROOT/scripts/gpu_quiet_probe.py 1.0 4 4096 random > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k in ('k_fused', 'k_quiet', 'k_step'):
            if k in r['Kernel_Name']:
                tot[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
res = {"%s.%s_KB_per_launch" % k: tot[k] / max(n[k], 1) for k in tot}
res['launches'] = {"%s.%s" % k: n[k] for k in n}
json.dump(res, open("$ROOT/gpurun_out/pmc_traffic_$TAG.json", "w"), indent=1)
print(res)
PY
