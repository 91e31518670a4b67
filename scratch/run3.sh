#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 600 python -m pytest tests -m gpu -x -q --timeout 200 --deselect "tests/test_gpu_parity.py::test_other_horizons_match_oracle" > $O/r2_pytest_$T.log 2>&1; tail -5 $O/r2_pytest_$T.log
timeout 300 python scratch/stamps2.py > $O/r2_stamps_$T.log 2>&1; cat $O/r2_stamps_$T.log
timeout 300 python scratch/ab.py libqtos_planner_r1.so libqtos_planner.so > $O/r2_ab_$T.log 2>&1; tail -4 $O/r2_ab_$T.log | grep kkt
