#!/bin/bash
# A/B of k_kkt2 against QTOS_KKT=$2 + parity subset + stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-i}; K=${2:-4}
AB_KKT=$K AB_GAITS=walk,trot timeout 400 python scratch/ab4.py > $O/r4_ab_$T.log 2>&1
QTOS_KKT=$K timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord or step_terrain or two_phase or factor or trot_gait" 2>&1 | tail -4 > $O/r4_tests_$T.log
QTOS_KKT=$K QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py > $O/r4_stamps_$T.log 2>&1
grep -v "^qtos:\|amdgpu.ids" $O/r4_ab_$T.log | tail -12; cat $O/r4_tests_$T.log; head -20 $O/r4_stamps_$T.log | cut -c1-110
