#!/bin/bash
# A/B of development libraries on one box: scratch/ab5.sh <kkt> lib1 lib2 ...   (QTOS_KKT=<kkt>; walk and trot, kkt launch ms, plans/s)
K=$1; shift
for L in "$@"; do
  for G in walk trot; do
    QTOS_KKT=$K QTOS_LIB=libqtos_$L.so timeout 300 python bench.py --steps 30 --cpu-sample 0 --no-parity --no-trot --gait $G 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-10s %-5s %-16s kkt %.4f ms  chord %.4f  plans/s %8.0f  ms/step %.3f conv %s/%s' % ('$L', '$G', r['kernel'], r['avg_launch_ms'], r.get('chord_avg_launch_ms') or 0, d['value'], d['ms_per_step'], d['config']['converged'], d['config']['plans_timed']))"
  done
done
