#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 900 python -m pytest tests -m gpu -x -q --timeout 240 -k "chord or golden or seeded" > $O/r2_pytest_$T.log 2>&1; tail -8 $O/r2_pytest_$T.log
timeout 600 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; cat $O/bench_$T.json; tail -3 $O/bench_$T.err
