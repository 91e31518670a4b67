#!/bin/bash
# final measurement pass of a round: bench lines, rocprofv3 kernel stats, PMC HBM traffic (separate passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
mkdir -p $O
cd $R
python bench.py > $O/bench_$T.json 2> $O/bench_$T.err
python bench.py --cpu-sample 0 --transcription reference_compat > $O/bench_${T}_compat.json 2>/dev/null
python bench.py --cpu-sample 0 --workload exp5_step > $O/bench_${T}_exp5.json 2>/dev/null
python bench.py --cpu-sample 0 --workload mixed > $O/bench_${T}_mixed.json 2>/dev/null
python bench.py --cpu-sample 0 --workload mixed --inflight 3 > $O/bench_${T}_mixed_inflight3.json 2>/dev/null
python bench.py --cpu-sample 0 --inflight 2 > $O/bench_${T}_flat_inflight2.json 2>/dev/null
python bench.py --cpu-sample 8 --transcription knots200 > $O/bench_${T}_knots200.json 2>/dev/null
python bench.py --cpu-sample 0 --transcription knots200 --workload mpc_random --steps 64 --max-iter 6 > $O/bench_${T}_mpc200_iter6.json 2>/dev/null
python bench.py --cpu-sample 0 --transcription knots200 --workload mpc_random --steps 64 > $O/bench_${T}_mpc200.json 2>/dev/null
python bench.py --cpu-sample 0 --init table > $O/bench_${T}_table.json 2>/dev/null
python bench.py --cpu-sample 0 --init table --transcription reference_compat > $O/bench_${T}_table_compat.json 2>/dev/null
python bench.py --cpu-sample 0 --init table --transcription knots200 > $O/bench_${T}_table_knots200.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o runc -- python3 $R/bench.py --cpu-sample 0 > $O/prof_$T.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/pmc_fetch_$T.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/pmc_write_$T.log 2>&1
cd $R; for f in $O/bench_$T*.json; do echo $f; tail -1 $f | cut -c1-160; done
find $O/prof_$T $O/pmc_fetch_$T $O/pmc_write_$T -name "*.csv" | head
