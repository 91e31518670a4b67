#!/bin/bash
# round 6: the order with the late force nodes (auto) against the order of rounds 1 - 5 (QTOS_ORDER=0) on one box: plans/s, launch time, converged
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
line() { python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']
    print('$1', j['value'], 'plans/s', j['ms_per_step'], 'ms/step', (j.get('roofline') or {}).get('kernel'), (j.get('roofline') or {}).get('avg_launch_ms'), 'front', c['front'], 'stages', c['kkt_stages'], 'converged', c['converged'], '/', c['plans_timed'], 'iters', c.get('iterations_mean'), c.get('replan_hz_per_window'))
except Exception as e: print('$1 FAILED', e)"; }
{
for o in auto 0; do
  if [ $o = auto ]; then unset QTOS_ORDER; else export QTOS_ORDER=$o; fi
  python bench.py $X --gait walk 2>/dev/null | line "order=$o walk"
  python bench.py $X --workload exp5_step 2>/dev/null | line "order=$o exp5"
  python bench.py $X --workload mixed 2>/dev/null | line "order=$o mixed"
  python bench.py $X --transcription knots200 2>/dev/null | line "order=$o knots200"
  python bench.py $X --transcription knots200 --workload mpc_random --steps 200 2>/dev/null | line "order=$o mpc"
  python bench.py $X --workload exp5_step --steps 300 2>/dev/null | line "order=$o exp5-300"
  python bench.py $X --workload mixed --steps 300 2>/dev/null | line "order=$o mixed-300"
done
} > $O/r6_order_bench.log 2>&1
cat $O/r6_order_bench.log
