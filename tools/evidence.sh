#!/bin/bash
# Runs on the GPU box: everything the round's profiles/ are made from, in order of importance.
#   tools/evidence.sh [quick]   quick: tests + smoke + the four bench lines only
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/round.sh
[ "$1" = quick ] && exit 0
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e" bash tools/prof.sh > gpurun_out/prof_console.log 2>&1; tail -5 gpurun_out/prof_console.log | cut -c1-600
bash tools/pipe_prof.sh 60 > gpurun_out/pipe_console.log 2>&1; tail -3 gpurun_out/pipe_console.log | cut -c1-400
bash tools/sq.sh "--config c4 --chroms 20 --steps 2 --warmup 1" c4 > gpurun_out/sq_c4_console.log 2>&1; tail -2 gpurun_out/sq_c4_console.log | cut -c1-600
bash tools/sq.sh "--config c3 --chroms 20 --steps 2 --warmup 1" c3 > gpurun_out/sq_c3_console.log 2>&1; tail -2 gpurun_out/sq_c3_console.log | cut -c1-600
