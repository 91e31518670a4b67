#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-p}; shift
AB_KKT=2 timeout 900 python scratch/ab4.py "$@" 2>&1 | grep -v "^qtos\|amdgpu" > $O/r4_ab_$T.log
cut -c1-150 $O/r4_ab_$T.log
