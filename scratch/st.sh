#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 300 python scratch/stamps2.py > $O/r3_stamps_$T.log 2>&1; cat $O/r2_stamps_$T.log
