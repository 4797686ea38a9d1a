#!/bin/bash
# SQ counters of k_step on mode 6 for several builds: scripts/pmc6.sh <tag> LIB1 LIB2 ...   (build_variants/lib_<LIB>.so)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  OUT=$ROOT/gpurun_out/pmc6_${TAG}_$L; mkdir -p $OUT; i=0
  export AGAR_LIB=$ROOT/build_variants/lib_$L.so
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_INSTS_SENDMSG"; do
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/scripts/pmc_run6.py 4096 ${PMC_STEPS:-240} > $OUT/p$i.log 2>&1
    i=$((i+1))
  done
  python3 - <<PY
import csv, glob, collections
vals = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if 'k_step' in r['Kernel_Name']]
    for r in rows: vals[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
print("== $L  (mean over the last 60 k_step dispatches: steady state)")
for k in sorted(vals):
    v = [x for _, x in sorted(vals[k])][-60:]
    print('%-24s per-dispatch %14.0f' % (k, sum(v) / len(v)))
PY
done
