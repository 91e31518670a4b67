#!/bin/bash
# round 6: bench lines of order bits 0 (rule 0) against 6 (early coefficients + first-knot guard) in the experiment library
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
for rep in 1 2; do
for b in 0 6; do
  for cfg in "--gait trot" "--transcription reference_compat --gait walk" "--transcription reference_compat --gait trot"; do
    QTOS_ORDER=0 QTOS_EXP_ORDERBITS=$b QTOS_LIB=libqtos_expo.so python bench.py $X $cfg 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
    print('bits %s %-45s %9.1f plans/s  kkt %.4f ms  chord %.4f ms  front %d  %s median step %.4f ms iters %s' % ('$b', '$cfg', j['value'], r['avg_launch_ms'], r['chord_avg_launch_ms'] or 0, j['config']['front'], r.get('kernel'), j['step_ms']['p50'], j.get('mean_iterations')))
except Exception as e: print('$b $cfg FAILED', e)"
  done
done
done
