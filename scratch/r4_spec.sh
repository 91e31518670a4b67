#!/bin/bash
# speculative Jacobian in k_step: A/B per workload + parity subset
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-x}
timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids > $O/r4_spec_ab_$T.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 > $O/r4_spec_tests_$T.log
cat $O/r4_spec_ab_$T.log $O/r4_spec_tests_$T.log
