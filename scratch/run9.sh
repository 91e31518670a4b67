#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1
timeout 600 python -m pytest tests -m gpu -x -q --timeout 200 --deselect "tests/test_gpu_parity.py::test_other_horizons_match_oracle" > $O/r2_pytest_$T.log 2>&1; tail -12 $O/r2_pytest_$T.log
timeout 300 python bench.py --cpu-sample 16 --no-parity > $O/r2_bench_$T.json 2>$O/r2_bench_$T.err; cut -c1-330 $O/r2_bench_$T.json; python - <<PY
import json; d=json.load(open("$O/r2_bench_$T.json")); print(d["roofline"]); print(d.get("cpu_baseline",{}).get("sample"))
PY
