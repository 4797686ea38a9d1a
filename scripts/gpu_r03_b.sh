#!/bin/bash
# round 3, GPU session B: the whole GPU suite, a 720-trial soak, the bench lines, the round's profiles
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
echo "== soak (8 procs x 90 trials)"; bash scripts/gpu_soak_par.sh 400 8 90 2>&1 | grep -E "^== seed|aperture|fault"
echo "== gpu_round"; bash scripts/gpu_round.sh r03b 2>&1 | tail -40
echo "== profile_round"; bash scripts/profile_round.sh r03b 2>&1 | tail -30
