#!/bin/bash
# The N > 1 path of bench.py on a 1-GPU box: 2 rank processes share GPU 0 over gloo (the driver's runs use nccl == RCCL);
# exercises the self-spawning launcher, the block / per-step result gathers and the observation gather.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/${1:-mr}; mkdir -p $O; cd $ROOT
export AGAR_BENCH_BACKEND=gloo
timeout 300 python bench.py --gpus 2 --steps 100 --warmup 20 > $O/g2_block.json 2> $O/g2_block.err; echo "rc=$?"
timeout 300 python bench.py --gpus 2 --steps 100 --warmup 20 --gather step > $O/g2_step.json 2> $O/g2_step.err; echo "rc=$?"
timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --gather step --gather-obs screen --arenas 1024 > $O/g2_obs.json 2> $O/g2_obs.err; echo "rc=$?"
WORLD_SIZE=3 timeout 60 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/mismatch.json 2> $O/mismatch.err; echo "mismatch rc=$? (must be non-zero)"
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/g2_*.json")):
    try:
        b=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), "n_gpus", b["n_gpus"], "%.4g env-steps/s  %.2f us/step" % (b["value"], b["ms_per_step"]*1e3), b["config"]["parallelism"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
