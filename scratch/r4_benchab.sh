#!/bin/bash
# default bench with the round's k_step switches off / on (alternating, 2 reps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
X="--cpu-sample 0 --no-parity --no-trot"
for rep in 1 2; do
for v in "0 0" "1 0" "0 1" "1 1"; do
set -- $v
QTOS_SWEEP_DS=$1 QTOS_SPEC_JAC=$2 python bench.py $X $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SWEEP_DS=$1 SPEC_JAC=$2 %9.0f plans/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
done; done
