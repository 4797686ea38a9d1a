#!/bin/bash
# SQ counters of k_screen_obs by ablation (build.py --variant SCRABL -DAG_SCR_ABL):  scripts/pmc_screen.sh <tag> <state> <W> <agent_view> <abl> [<abl> ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; ST=$2; W=$3; AV=$4; shift 4
export AGARCL_HIP_SO=$ROOT/build_variants/lib_SCRABL.so
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcs_${TAG}; mkdir -p $OUT
for ab in "$@"; do
  export AGARCL_SCR_ABL=$ab
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/a$ab -o pmc -- python3 $ROOT/scripts/pmc_screen.py $ST $W $AV > $OUT/a$ab.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/a*/")):
    vals = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_screen_obs" in r["Kernel_Name"]: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.rstrip("/").split("/")[-1], " ".join("%s=%.3g" % (k.replace("SQ_", ""), sum(v) / len(v)) for k, v in sorted(vals.items())))
PY
