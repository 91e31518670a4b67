#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-x}
AB_WLS=walk AB_VAR=QTOS_SWEEP_DS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids > $O/r4_sw2_ab_$T.log
for v in 0 1; do
QTOS_SWEEP_DS=$v QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py 2>&1 | grep -v amdgpu.ids | grep -E "wave  0|wave 13|k_step|helper turn" | cut -c1-330 >> $O/r4_sw2_ab_$T.log
done
cat $O/r4_sw2_ab_$T.log
