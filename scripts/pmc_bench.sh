#!/bin/bash
# SQ counters of the step kernels for a bench workload: scripts/pmc_bench.sh <tag> <workload> <arenas> [steps] [ENV=VAL ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; W=$2; A=$3; S=${4:-200}; shift 4
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcb_${TAG}; mkdir -p $OUT; i=0
WARM=40; [ "$W" = "mid" ] && WARM=400
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/bench.py --workload $W --arenas $A --steps $S --warmup $WARM --no-cpu-baseline --no-large --no-full > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = [n for n in ("k_fused", "k_quiet", "k_step") if n in r['Kernel_Name']]
        if k: vals[k[0]][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for kn in vals:
    print("== $W@$A  %s  (mean over the last %d dispatches)" % (kn, min($S, 100)))
    for c in sorted(vals[kn]):
        v = [x for _, x in sorted(vals[kn][c])][-min($S, 100):]
        print('%-24s per-dispatch %14.0f' % (c, sum(v) / len(v)))
PY
