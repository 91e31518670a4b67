#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-trot"
for rep in 1 2; do for v in 0 1; do for a in "" "--tol 1e-3" "--workload exp5_step"; do
QTOS_SPEC_JAC=$v python bench.py $X $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SPEC_JAC=$v %-22s %9.0f plans/s %.4f ms/step' % ('$a', d['value'], d['ms_per_step']))"
done; done; done
