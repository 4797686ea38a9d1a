#!/bin/bash
# The round's GPU session in one call: the GPU suite, the bench line as the driver runs it and at its defaults, the rocprofv3 passes of
# scripts/profile_round.sh (kernel trace, FETCH_SIZE / WRITE_SIZE, one SQ pass -> gpurun_out/profiles_<tag>/) and a parallel soak.
#   scripts/gpu_session_r05.sh <tag> [soak procs] [soak trials]
TAG=${1:-r05}; PROCS=${2:-14}; TRIALS=${3:-150}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver20.json 2> $O/bench_driver20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash scripts/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -12 $O/profile_round.log
bash scripts/gpu_soak_par.sh 900 $PROCS $TRIALS > $O/soak_default.log 2>&1; tail -$((PROCS + 2)) $O/soak_default.log | cut -c1-200
bash scripts/gpu_soak_par.sh 950 $PROCS $((TRIALS / 3)) SOAK_MANY=1 > $O/soak_many.log 2>&1; tail -$((PROCS + 2)) $O/soak_many.log | cut -c1-200
cat gpurun_out/soak_9*.log | grep -E "flagged \(capacity\)" | sed 's/.*flags //' | sort | uniq -c > $O/soak_flag_histogram.txt; cat $O/soak_flag_histogram.txt
rm -f gpurun_out/soak_9*.log
