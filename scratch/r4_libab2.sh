#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
LIBS=${LIBS:-libqtos_planner_old.so,libqtos_planner.so}
for cfg in knots100 reference_compat knots200; do
AB_CFG=$cfg AB_WLS=walk AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$cfg /" | sed 's/; conv.*sha/ sha/'
done
AB_WLS=trot AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/'
