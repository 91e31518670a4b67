#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/r4_final_tests.log
timeout 300 python __graft_entry__.py smoke > $O/r4_final_smoke.log 2>&1
timeout 900 python bench.py > $O/r4_final_bench.json 2> $O/r4_final_bench.err
cat $O/r4_final_tests.log; tail -2 $O/r4_final_smoke.log; python - <<PY
import json
d = json.loads(open("$O/r4_final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["trot"]["value"], d["roofline"]["traffic_source"], d["roofline"]["counters"]["source"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
