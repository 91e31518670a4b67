#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 600 python -m pytest tests -m gpu -x -q --timeout 200 > $O/r2_pytest_$T.log 2>&1; tail -8 $O/r2_pytest_$T.log
timeout 600 python bench.py > $O/r2_bench_$T.json 2> $O/r2_bench_$T.err; cut -c1-400 $O/r2_bench_$T.json; tail -3 $O/r2_bench_$T.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --cpu-sample 0 --no-parity > $O/r2_bench_${T}_dist1.json 2> $O/r2_bench_${T}_dist1.err; cut -c1-300 $O/r2_bench_${T}_dist1.json; tail -3 $O/r2_bench_${T}_dist1.err
timeout 600 python bench.py --cpu-sample 0 --no-parity --workload mixed > $O/r2_bench_${T}_mixed.json 2>/dev/null; cut -c1-200 $O/r2_bench_${T}_mixed.json
timeout 600 python bench.py --cpu-sample 0 --no-parity --gait trot > $O/r2_bench_${T}_trot.json 2>/dev/null; cut -c1-200 $O/r2_bench_${T}_trot.json
