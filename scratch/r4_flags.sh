#!/bin/bash
# compiler-flag variants of the whole library against the default build (csrc/libqtos_planner_v*.so)
R=$GRAFT_REPO_ROOT; cd $R
LIBS=${LIBS:-libqtos_planner.so,libqtos_planner_v1.so,libqtos_planner_v2.so,libqtos_planner_v4.so,libqtos_planner_v5.so,libqtos_planner_v6.so,libqtos_planner_v7.so}
for wl in walk trot; do
AB_WLS=$wl AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 1200 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/' | grep -v "max diff"
done
