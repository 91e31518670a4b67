#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 900 python -m pytest tests -m gpu -x -q --timeout 240 --durations=8 > $O/r2_pytest_$T.log 2>&1; tail -16 $O/r2_pytest_$T.log
