#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "replan_loop_state or shifted_windows" > $O/r2_pytest_$T.log 2>&1; tail -30 $O/r2_pytest_$T.log
timeout 600 python bench.py --cpu-sample 0 --no-parity --transcription knots200 --workload mpc_random --steps 200 --warmup 3 > $O/r2_bench_${T}_mpc.json 2> $O/r2_bench_${T}_mpc.err; cut -c1-1500 $O/r2_bench_${T}_mpc.json; tail -5 $O/r2_bench_${T}_mpc.err
