#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/r2_pytest_b.log 2>&1; tail -15 $O/r2_pytest_b.log
timeout 300 python scratch/ab.py libqtos_planner_r1.so libqtos_planner.so > $O/r2_ab_b.log 2>&1; cat $O/r2_ab_b.log
