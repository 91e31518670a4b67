#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
timeout 900 python scratch/ab3.py "$@" > $O/r3_ab3_$T.log 2>&1; grep -E "kkt|Error|error" $O/r3_ab3_$T.log | sort
timeout 300 python scratch/stamps2.py > $O/r3_stamps_$T.log 2>&1; tail -4 $O/r3_stamps_$T.log
rm -f $O/ab3_*.npy
