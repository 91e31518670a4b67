#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pool_of_handles or knots100_batch or golden_inputs or chord or full_batch or shifted_windows_match or other_horizons" --durations=8 2>&1 | tail -25
for w in "" "--workload exp5_step" "--workload mixed" "--workload exp5_step --inflight 3" "--workload mixed --inflight 3" "--inflight 2"; do
  echo "== bench $w"; timeout 600 python bench.py $w --cpu-sample 0 --no-parity --no-trot > $O/tmp_b.log 2>&1; tail -1 $O/tmp_b.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'].get('converged'), d.get('roofline', {}).get('avg_launch_ms'), d.get('roofline', {}).get('launches_per_step'))
except Exception as e: print('ERR', e); print(open('$O/tmp_b.log').read()[-1500:])
"
done
echo "== mpc"; timeout 900 python bench.py --transcription knots200 --workload mpc_random --steps 60 --cpu-sample 0 --no-parity > $O/tmp_b.log 2>&1; tail -1 $O/tmp_b.log | cut -c1-400
echo "== mpc one set"; timeout 900 python bench.py --transcription knots200 --workload mpc_random --steps 60 --inflight 1 --cpu-sample 0 --no-parity > $O/tmp_b.log 2>&1; tail -1 $O/tmp_b.log | cut -c1-400
