#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
for n in 1 2 4; do
timeout 600 python bench.py --cpu-sample 0 --no-parity --transcription knots200 --workload mpc_random --steps 200 --warmup 3 --inflight $n > $O/r2_bench_${T}_mpc_if$n.json 2> $O/r2_bench_${T}_mpc.err; python - <<PY
import json; d=json.load(open("$O/r2_bench_${T}_mpc_if$n.json")); print("inflight $n:", d["value"], d["ms_per_step"], d["config"]["replan_hz_per_window"], d["config"]["converged_fraction"], d["config"]["iterations_mean"])
PY
done
tail -3 $O/r2_bench_${T}_mpc.err
