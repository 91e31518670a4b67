#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-trot --transcription knots200 --workload mpc_random --steps 100"
for a in "--inflight 4" "--inflight 8" "--inflight 2" "--inflight 16"; do
  python bench.py $X $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %9.0f plans/s %7.3f ms/step conv %s/%s' % ('$a', d['value'], d['ms_per_step'], d['config'].get('converged'), d['config'].get('plans_timed')))"
done
