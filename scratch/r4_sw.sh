#!/bin/bash
# ds = Ji dx on the helper waves of the backward sweep: A/B per workload (+ other transcriptions) + parity tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-x}
AB_VAR=QTOS_SWEEP_DS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids > $O/r4_sw_ab_$T.log
cat $O/r4_sw_ab_$T.log
if [ "$2" != "quick" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r4_sw_tests_$T.log
cat $O/r4_sw_tests_$T.log
fi
