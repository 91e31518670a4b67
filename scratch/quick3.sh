#!/bin/bash
# quick look at the workloads whose iteration pattern a solver change can move: trot, exp_5, mixed, receding windows
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-trot"
for a in "" "--gait trot" "--workload exp5_step" "--workload mixed" "--transcription knots200 --workload mpc_random --steps 100" "--transcription knots200" "--transcription reference_compat"; do
  python bench.py $X $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('%-60s %9.0f plans/s %7.3f ms/step  kkt/step %s chord/step %s conv %s/%s' % ('$a', d['value'], d['ms_per_step'], r.get('launches_per_step'), r.get('chord_launches_per_step'), d['config'].get('converged'), d['config'].get('plans_timed')))"
done
