#!/bin/bash
# One GPU session of a round, in one gpurun call:  scripts/gpu_session.sh <tag> [soak procs] [soak trials per proc]
#   1. the GPU test suite                                   -> gpurun_out/<tag>/pytest.log
#   2. the bench line as the driver runs it, and at defaults -> gpurun_out/<tag>/bench_driver20.json (+ _full.json), bench_default.json
#   3. scripts/profile_round.sh <tag>                        -> gpurun_out/profiles_<tag>/  (kernel stats, PMC traffic, SQ issue figures)
#   4. a parallel soak on the default distribution (+ SOAK_MANY): scripts/gpu_soak_par.sh -> gpurun_out/<tag>/soak.txt
# then, in the build container:  scripts/adopt_profiles.sh <tag> r06 --resources
TAG=${1:-r06}; PROCS=${2:-8}; TRIALS=${3:-250}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver20.json 2> $O/bench_driver20.err; cp bench_full.json $O/bench_driver20_full.json; wc -c $O/bench_driver20.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cp bench_full.json $O/bench_default_full.json
bash scripts/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
cd $ROOT
bash scripts/gpu_soak_par.sh 1200 $PROCS $TRIALS > $O/soak.txt 2>&1
bash scripts/gpu_soak_par.sh 1300 $PROCS $((TRIALS / 3)) SOAK_MANY=1 >> $O/soak.txt 2>&1
grep "soak done" $O/soak.txt | tail -40
