#!/bin/bash
# Per-launch duration of the step kernel over the first 120 steps after a reset (rocprofv3 kernel trace): how the early episode differs from the steady state.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/trend
rocprofv3 --kernel-trace --output-format csv -d /tmp/trend -o t -- python3 $ROOT/bench.py --steps 115 --warmup 5 --no-cpu-baseline --no-large > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/trend/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; duration (us) of steps 0..119 in tens:")
for i in range(0, len(d), 10): print("%3d: " % i + " ".join("%5.2f" % x for x in d[i:i + 10]))
PY
