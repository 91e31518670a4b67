#!/bin/bash
# round 6: library builds against each other on more configurations (KKT ms per launch is the number: plans/s on a box with host stalls is noisy)
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
for rep in 1 2; do
for n in "$@"; do
  for cfg in "--gait trot" "--gait walk" "--transcription reference_compat --gait walk" "--transcription reference_compat --gait trot" "--transcription knots200" "--workload exp5_step" "--workload mixed"; do
    QTOS_LIB=libqtos_$n.so python bench.py $X $cfg 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
    print('%-10s %-45s %9.1f plans/s  kkt %.4f ms  chord %.4f ms  front %d  median step %.4f ms' % ('$n', '$cfg', j['value'], r['avg_launch_ms'], r['chord_avg_launch_ms'] or 0, j['config']['front'], j['step_ms']['p50']))
except Exception as e: print('$n $cfg FAILED', e)"
  done
done
done
