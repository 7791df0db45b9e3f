#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel + memory-copy trace of the end-to-end (drop-in) leg,
# summarised into gpurun_out/pipe/overlap.json (copy / compute overlap of the streaming pipeline).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pipe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/p_pipe -- python $R/tools/e2e_only.py ${1:-60} > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-1500
python $R/tools/pipe_overlap.py /tmp/p_pipe $OUT/overlap.json
