#!/bin/bash
# one full GPU session of a round (soak on the final code, GPU suite, bench lines, profiles): run through gpurun, then scripts/adopt_profiles.sh <tag>; the whole GPU suite, a soak on the final code, the bench lines, the round's profiles
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
echo "== soak (8 procs x 40 trials)"; bash scripts/gpu_soak_par.sh 500 8 40 2>&1 | grep -E "^== seed|aperture|fault"; rm -f gpurun_out/soak_*.log
echo "== gpu_round"; bash scripts/gpu_round.sh ${1:-r03d} 2>&1 | tail -32
echo "== C1r"; python bench.py --workload C1r --steps 200 --warmup 40 --no-cpu-baseline --no-full 2>/dev/null | python3 -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C1r', b['value'], b['ms_per_step']*1e3, 'us')"
echo "== obs timing"; timeout 300 python scripts/gpu_obs_timing.py 2>&1 | tail -6
echo "== profile_round"; bash scripts/profile_round.sh ${1:-r03d} 2>&1 | tail -12
du -sh gpurun_out
