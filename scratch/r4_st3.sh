#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
QTOS_SWEEP_DS=0 QTOS_LIB=libqtos_planner_exp.so timeout 300 python scratch/stamps2.py 2>&1 | grep -v amdgpu.ids | grep -E "wave  0|chain wave|work before" | cut -c1-250
QTOS_SWEEP_DS=0 QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py 2>&1 | grep -v amdgpu.ids | grep -E "wave  0|chain wave|work before" | cut -c1-250
