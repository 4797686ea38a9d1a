#!/bin/bash
# round 3, GPU session C: new tests, bench lines, the round's profiles
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_ram_obs.py tests/test_video_glue.py tests/test_000_bench_ranks_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== gpu_round"; bash scripts/gpu_round.sh r03c 2>&1 | tail -30
echo "== C1r"; python bench.py --workload C1r --steps 200 --warmup 40 --no-cpu-baseline --no-full 2>/dev/null | python3 -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C1r', b['value'], b['ms_per_step']*1e3, 'us')"
echo "== profile_round"; bash scripts/profile_round.sh r03c 2>&1 | tail -12
du -sh gpurun_out
