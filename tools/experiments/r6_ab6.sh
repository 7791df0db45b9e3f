#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "difference_array or exact or golden or config_sized" 2>&1 | tail -1
tools/r6_ab.sh ab6 - ze "c2 0 16" "c2 20 200" "c2 0 64"
bash tools/experiments/r6_prof_ab.sh prof
