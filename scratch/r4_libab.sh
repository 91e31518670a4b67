#!/bin/bash
# A/B of libraries built with different sweep variants (csrc/libqtos_planner_*.so), ds in the sweep off and on
R=$GRAFT_REPO_ROOT; cd $R
LIBS=${LIBS:-libqtos_planner_old.so,libqtos_planner.so}
for ds in 0 1; do
QTOS_SWEEP_DS=$ds AB_WLS=${AB_WLS:-walk} AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed "s/^/DS=$ds /" | sed 's/; conv.*sha/ sha/'
done
