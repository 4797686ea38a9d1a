#!/bin/bash
# HBM traffic of k_step from the TCC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one
# pass on gfx950), counters only (no tracing domains).  Output: gpurun_out/pmc_traffic_<tag>.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$ROOT/gpurun_out/traffic_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- python3 $ROOT/scripts/pmc_run.py 4096 4 > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
res = {k: tot[k] / max(n[k], 1) for k in tot}
res['launches'] = {k: n[k] for k in n}
json.dump(res, open("$ROOT/gpurun_out/pmc_traffic_$TAG.json", "w"))
print(res)
PY
