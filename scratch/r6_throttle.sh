#!/bin/bash
# round 6: host stalls of 9 - 11 ms in the timed loop -- CFS throttling of the container by spinning OpenMP / MKL helper threads?
# bench lines with the anti-spin environment bench.py sets by default against the old behaviour, one box; step-time maxima,
# host time outside the device time, throttled ms of the cgroup during the timed steps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait --steps 300"
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
big=[x for x in j.get('step_ms_series',[]) if x>4]
print('$1', j['value'], 'plans/s p50', j['step_ms']['p50'], 'max', j['step_ms']['max'], 'host_out', j['host_ms_per_step_outside_device_time'], 'throttled ms', j.get('cgroup_throttled_ms'), 'periods', j.get('cgroup_throttled_periods'), 'gap', j['gap_ms_per_step'])"; }
{
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for rep in 1 2 3; do
  python bench.py $X 2>/dev/null | line "default-env   "
  OMP_WAIT_POLICY=active KMP_BLOCKTIME=200 GOMP_SPINCOUNT=300000 MKL_NUM_THREADS=256 OPENBLAS_NUM_THREADS=256 python bench.py $X 2>/dev/null | line "spinning-env  "
done
} > $O/r6_throttle.log 2>&1
cat $O/r6_throttle.log
