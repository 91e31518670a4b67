#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-a}
QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py > $O/r4_stamps_$T.log 2>&1
head -20 $O/r4_stamps_$T.log
