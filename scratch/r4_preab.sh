#!/bin/bash
# factorisation queued ahead of the counts (QTOS_PRE_KKT): bench A/B over workloads + full GPU tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
X="--cpu-sample 0 --no-parity --no-trot"
for rep in 1 2; do for a in "" "--gait trot" "--workload exp5_step" "--workload mixed" "--inflight 2" "--workload exp5_step --inflight 3" "--transcription knots200"; do for v in 0 1; do
QTOS_PRE_KKT=$v python bench.py $X $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRE_KKT=$v %-36s %9.0f plans/s %.4f ms/step conv %s' % ('$a', d['value'], d['ms_per_step'], d['config'].get('converged')))"
done; done; done | tee $O/r4_preab.log
if [ "$1" != "quick" ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; fi
