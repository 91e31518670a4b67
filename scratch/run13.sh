#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
b() { name=$1; shift; timeout 900 python bench.py --cpu-sample 0 --no-parity "$@" > $O/bench_${T}_$name.json 2> $O/bench_${T}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${T}_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline",{}); print("$name", d["value"], d["unit"], d["ms_per_step"], "ms/step", d["config"].get("replan_hz_per_window"), "Hz", d["config"].get("converged"), "/", d["config"].get("plans_timed"))
except Exception as e: print("$name FAILED", e); print(open("$O/bench_${T}_$name.err").read()[-600:])
PY
}
b mpc2 --transcription knots200 --workload mpc_random --steps 100 --inflight 2
b mpc4 --transcription knots200 --workload mpc_random --steps 100 --inflight 4
b mpc8 --transcription knots200 --workload mpc_random --steps 100 --inflight 8
b mpc4_nochord --transcription knots200 --workload mpc_random --steps 100 --inflight 4 --chord-tol 0
b mpc8_nochord --transcription knots200 --workload mpc_random --steps 100 --inflight 8 --chord-tol 0
