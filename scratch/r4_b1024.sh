#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-a}
for w in exp5_step mixed exp1_flat; do
  for b in 256 1024; do
    timeout 600 python bench.py --workload $w --batch $b --steps 40 --warmup 4 --cpu-sample 0 --no-parity --no-trot > $O/r4_b_${w}_${b}_$T.json 2> $O/r4_b_${w}_${b}_$T.err
    python - <<PY
import json
d = json.loads(open("$O/r4_b_${w}_${b}_$T.json").read().strip().splitlines()[-1])
print("$w", $b, d["value"], d["ms_per_step"], d["config"].get("converged"), d["config"].get("plans_timed"))
PY
  done
done
