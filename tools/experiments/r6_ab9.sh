#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
tools/r6_ab.sh ab9 nohc - "c2 0 16" "c2 20 200" "c2 0,1,2,20,21 200" "c3 0 16"
