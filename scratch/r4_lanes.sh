#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-a}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes or pool_of_handles or knots100_batch or shifted_windows" 2>&1 | tail -6 > $O/r4_lanes_tests_$T.log
cat $O/r4_lanes_tests_$T.log
for w in exp5_step mixed exp1_flat; do
  for l in 4 2 1; do
    QTOS_LANES=$l timeout 600 python bench.py --workload $w --batch 1024 --steps 40 --warmup 4 --cpu-sample 0 --no-parity --no-trot > $O/r4_l_${w}_${l}_$T.json 2> $O/r4_l_${w}_${l}_$T.err
    python - <<PY
import json
d = json.loads(open("$O/r4_l_${w}_${l}_$T.json").read().strip().splitlines()[-1])
print("$w lanes $l:", d["value"], d["ms_per_step"], d["config"].get("converged"), d["config"].get("plans_timed"))
PY
  done
done
