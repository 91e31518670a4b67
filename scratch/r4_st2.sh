#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in 0 1; do
QTOS_SWEEP_DS=$v QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py 2>&1 | grep -v amdgpu.ids | grep -E "wave  0|wave 13|k_step|helper turn|chain wave" | cut -c1-250
done
