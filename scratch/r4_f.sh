#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-m}
(AB_KKT=4 timeout 400 python scratch/ab4.py; QTOS_NO_SHORT_STAGES=1 AB_KKT=4 timeout 400 python scratch/ab4.py) 2>&1 | grep -v "^qtos\|amdgpu" > $O/r4_ab_$T.log
cut -c1-140 $O/r4_ab_$T.log
