#!/bin/bash
# A/B (walk, trot) + parity subset + stamps in one call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-b}
QTOS_DEBUG_SYMBOLIC=1 AB_GAITS=walk,trot timeout 400 python scratch/ab4.py > $O/r4_ab_$T.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord or step_terrain or two_phase or factor or trot" 2>&1 | tail -8 > $O/r4_tests_$T.log
QTOS_LIB=libqtos_planner_stamps.so timeout 300 python scratch/stamps2.py > $O/r4_stamps_$T.log 2>&1
grep -v "^qtos:\|amdgpu.ids" $O/r4_ab_$T.log | tail -20; grep "qtos: k_kkt3\|LDS" $O/r4_ab_$T.log | sort | uniq -c; cat $O/r4_tests_$T.log; head -20 $O/r4_stamps_$T.log | cut -c1-120
