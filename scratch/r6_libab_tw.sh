#!/bin/bash
# round 6: library builds against each other on the two benchmark gaits (KKT ms per launch), three alternating passes; usage: r6_libab_tw.sh name...
# (csrc/libqtos_<name>.so; "planner" = the product library)
R=$GRAFT_REPO_ROOT; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
for rep in 1 2 3; do
for n in "$@"; do
  for cfg in "--gait trot" "--gait walk"; do
    QTOS_LIB=libqtos_$n.so python bench.py $X $cfg 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
    print('%-10s %-14s %9.1f plans/s  kkt %.4f ms  chord %.4f ms  median step %.4f ms' % ('$n', '$cfg', j['value'], r['avg_launch_ms'], r['chord_avg_launch_ms'] or 0, j['step_ms']['p50']))
except Exception as e: print('$n $cfg FAILED', e)"
  done
done
done
python scratch/ab_hash.py $(for n in "$@"; do echo libqtos_$n.so; done) 2>&1 | grep -v amdgpu.ids | tail -12
