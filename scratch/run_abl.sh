#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
timeout 600 python scratch/abl.py "$@" > $O/r3_abl_$T.log 2>&1
timeout 300 python scratch/stamps2.py > $O/r3_stamps_$T.log 2>&1
cat $O/r3_abl_$T.log | grep kkt; cat $O/r3_stamps_$T.log | tail -30
