#!/bin/bash
# rocprofv3 kernel stats of the default bench (no counters): per-kernel average durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o runc -- python3 $R/bench.py --cpu-sample 0 --no-parity --no-trot > $O/prof_$T.log 2>&1
tail -1 $O/prof_$T.log | cut -c1-300
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
