#!/bin/bash
# round 3, GPU call A: the fix against the recorded fault, the aperture microbenchmark, the GPU test suite, a parallel soak, phase profiles
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT 2>/dev/null || true
echo "== aperture microbench"; ./build_variants/flat_lds_aperture 0; ./build_variants/flat_lds_aperture 1 2>&1 | tail -2
echo "== replay, round-2 kernel (BASE) vs fixed (FIX)"; python scripts/gpu_fused_fault.py 2 BASE,FIX 2>&1 | grep -E "^variant"
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
echo "== soak (8 procs x 40 trials)"; bash scripts/gpu_soak_par.sh 300 8 40
echo "== phase profile mode 6 / C1"; timeout 300 python scripts/gpu_phase6.py 2>&1 | tail -45
echo "== grown"; timeout 300 python scripts/gpu_grown.py 2>&1 | tail -6
