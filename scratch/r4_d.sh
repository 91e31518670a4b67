#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-k}
for c in reference_compat knots200; do AB_CFG=$c AB_KKT=4 AB_GAITS=walk,trot timeout 600 python scratch/ab4.py 2>&1 | grep -v "^qtos\|amdgpu"; done > $O/r4_ab_$T.log 2>&1
AB_KKT=4 AB_GAITS=trot timeout 300 python scratch/ab4.py 2>&1 | grep -v "^qtos\|amdgpu" >> $O/r4_ab_$T.log
cat $O/r4_ab_$T.log | cut -c1-150
