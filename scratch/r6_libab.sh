#!/bin/bash
# round 6: A/B of library builds (csrc/libqtos_<name>.so) on one box: trot and walk, KKT ms per launch and plans/s; usage: r6_libab.sh name...
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
for rep in 1 2; do
for n in "$@"; do
  for g in trot walk; do
    QTOS_LIB=libqtos_$n.so python bench.py $X --gait $g 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
    print('%-10s %-5s %9.1f plans/s  kkt %.4f ms  chord %.4f ms  step+counts %.4f  front %d  converged %d' % ('$n', '$g', j['value'], r['avg_launch_ms'], r['chord_avg_launch_ms'], j['kernel_ms_per_step']['k_step_and_counts'], j['config']['front'], j['config']['converged']))
except Exception as e: print('$n $g FAILED', e)"
  done
done
done
