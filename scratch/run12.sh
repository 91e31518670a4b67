#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
b() { name=$1; shift; timeout 900 python bench.py --cpu-sample 0 --no-parity "$@" > $O/bench_${T}_$name.json 2> $O/bench_${T}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${T}_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline",{}); print("$name", d["value"], d["unit"], d["ms_per_step"], "ms/step kkt", r.get("avg_launch_ms"), "x", r.get("launches_per_step"), "chord", r.get("chord_avg_launch_ms"), "x", r.get("chord_launches_per_step"), d["config"].get("converged"), "/", d["config"].get("plans_timed"), "it mean", d["config"].get("iterations_mean"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
b mixed --workload mixed
b exp5 --workload exp5_step
b mpc --transcription knots200 --workload mpc_random --steps 100
timeout 900 python -m pytest tests -m gpu -x -q --timeout 900 > $O/r2_pytest_$T.log 2>&1; tail -3 $O/r2_pytest_$T.log
