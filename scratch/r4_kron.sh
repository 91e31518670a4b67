#!/bin/bash
# Kronecker assembly of the range-of-motion blocks (QTOS_KRON=1, k_kkt2<128, false, true>): parity subset + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
QTOS_KRON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord or step_terrain or two_phase or factor" 2>&1 | tail -5
AB_WLS=${AB_WLS:-walk,exp5} AB_VAR=QTOS_KRON timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv/ conv/'
