#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { echo "== $*"; timeout 900 env $ENVV python bench.py "$@" --cpu-sample 0 --no-parity --no-trot > $O/tmp_b.log 2>&1; tail -1 $O/tmp_b.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); c = d['config']; print(d['value'], d['ms_per_step'], c.get('converged'), c.get('plans_timed'), d.get('roofline', {}).get('avg_launch_ms'), d.get('roofline', {}).get('launches_per_step'), c.get('replan_hz_per_window'))
except Exception as e: print('ERR', e); print(open('$O/tmp_b.log').read()[-1500:])
"; }
for cap in 1 8; do
export ENVV="QTOS_SPEC_CAP=$cap"; echo "#### cap $cap"
run
run --workload exp5_step
run --workload mixed
run --workload exp5_step --inflight 3
run --transcription knots200 --workload mpc_random --steps 60
done
