#!/bin/bash
# full GPU test suite + default bench line (+ a k_kkt3 parity subset through QTOS_KKT=3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-x}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/r4_full_tests_$T.log
QTOS_KKT=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord or step_terrain or two_phase or factor or trot_gait" 2>&1 | tail -4 > $O/r4_full_kkt3_$T.log
timeout 900 python bench.py > $O/r4_full_bench_$T.json 2> $O/r4_full_bench_$T.err
cat $O/r4_full_tests_$T.log $O/r4_full_kkt3_$T.log; python - <<PY
import json
d = json.loads(open("$O/r4_full_bench_$T.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d.get("trot", {}).get("value"), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print(d["cpu_baseline"])
PY
