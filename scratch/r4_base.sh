#!/bin/bash
# round 4 baseline: GPU tests, default bench line, stamps of the unchanged kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/r4_base_tests.log
timeout 600 python bench.py > $O/r4_base_bench.json 2> $O/r4_base_bench.err
QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py > $O/r4_base_stamps.log 2>&1
cat $O/r4_base_tests.log; cut -c1-600 $O/r4_base_bench.json; head -20 $O/r4_base_stamps.log
