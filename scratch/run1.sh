#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/r2_pytest_a.log 2>&1; tail -3 $O/r2_pytest_a.log
python scratch/ab.py libqtos_planner_r1.so libqtos_planner.so > $O/r2_ab_a.log 2>&1; cat $O/r2_ab_a.log
./scratch/ub/ub_prims > $O/r2_ub_prims.log 2>&1; tail -30 $O/r2_ub_prims.log
