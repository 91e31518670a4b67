#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for lib in libqtos_planner.so libqtos_planner_a0.so libqtos_planner_f2.so; do
QTOS_LIB=$lib AB_WLS=walk AB_VAR=QTOS_KRON timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/' | grep -v "max diff" | sed "s/^/$lib /"
done
