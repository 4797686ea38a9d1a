#!/bin/bash
# rocprofv3 PMC passes (counters only: never combined with sys/hip/hsa tracing) on scripts/pmc_run.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS"; do
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/scripts/pmc_run.py "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print('%-24s per-dispatch %14.0f' % (k, tot[k] / max(n[k], 1)))
PY
