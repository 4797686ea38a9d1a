#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for lib in $ROOT/agarcl_amd/libabl_*.so; do
  OUT=$ROOT/gpurun_out/abl_$(basename $lib .so); mkdir -p $OUT
  AGAR_LIB=$lib timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_VMEM_RD --output-format csv -d $OUT -o pmc -- python3 $ROOT/scripts/pmc_run.py 4096 ${1:-4} > $OUT.log 2>&1
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
w = tot['SQ_WAVES'] / max(n['SQ_WAVES'], 1)
print("$(basename $lib)", ' '.join('%s=%.0f' % (k.replace('SQ_INSTS_', ''), tot[k] / max(n[k], 1) / w) for k in sorted(tot) if k != 'SQ_WAVES'), '(per wave per launch)')
PY
done
