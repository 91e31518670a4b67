#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1; do
QTOS_KRON=$v QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py 2>&1 | grep -v amdgpu.ids | grep -E "^wave|stage total" | cut -c1-120
done
