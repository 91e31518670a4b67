#!/bin/bash
# longer runs of the workloads (300 fresh batches each): every plan converged?
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-trot --steps 300"
for a in "--gait trot" "--workload exp5_step" "--workload mixed" "--transcription knots200" "--transcription reference_compat --workload exp5_step" "--transcription reference_compat --gait trot" "--gait walk"; do
  python bench.py $X $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-55s %9.0f plans/s conv %s/%s it %s' % ('$a', d['value'], d['config'].get('converged'), d['config'].get('plans_timed'), d['config'].get('iterations_mean')))"
done
