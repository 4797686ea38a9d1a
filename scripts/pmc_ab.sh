#!/bin/bash
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: separate passes, counters only) of the step kernels for several BUILDS of the library on one
# bench workload:  scripts/pmc_ab.sh <tag> <workload> <arenas> <steps> LIB1 LIB2 ...   (LIB = product | build_variants/lib_<LIB>.so)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; W=$2; A=$3; S=$4; shift 4
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcab_${TAG}; mkdir -p $OUT
WARM=40; [ "$W" = "mid" ] && WARM=400
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset AGARCL_HIP_SO; else export AGARCL_HIP_SO=$ROOT/build_variants/lib_$lib.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/${lib}_$c -o pmc -- python3 $ROOT/bench.py --workload $W --arenas $A --steps $S --warmup $WARM --no-cpu-baseline --no-large --no-full > $OUT/${lib}_$c.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections, os
for lib in "$*".split():
    tot = collections.defaultdict(list)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/%s_%s/*counter_collection.csv" % (lib, c)):
            for r in csv.DictReader(open(f)):
                k = [n for n in ("k_fused", "k_quiet", "k_step", "k_grid_obs", "k_screen_obs") if n in r['Kernel_Name']]
                if k: tot[(k[0], r['Counter_Name'])].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for (k, c), v in sorted(tot.items()):
        v = [x for _, x in sorted(v)][-min($S, 100):]
        kb = sum(v) / len(v)
        print("%-10s $W@$A %-12s %-10s %10.1f KB per launch%s" % (lib, k, c, kb, "  (x2 on gfx950 for bytes: %.2f MB)" % (2 * kb / 1024) if c == "FETCH_SIZE" else "  = %.2f MB" % (kb / 1024)))
PY
rm -rf $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE
