#!/bin/bash
# round 4: first run of k_kkt3 -- a parity subset, then A/B against k_kkt2 in the same library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-a}
QTOS_DEBUG_SYMBOLIC=1 timeout 300 python scratch/ab4.py > $O/r4_ab_$T.log 2>&1
AB_GAITS=trot timeout 300 python scratch/ab4.py >> $O/r4_ab_$T.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord or step_terrain or two_phase or pool_of_handles or factor" 2>&1 | tail -8 > $O/r4_tests_$T.log
grep -v "^qtos:\|amdgpu.ids" $O/r4_ab_$T.log | tail -20; grep "qtos: k_kkt3" $O/r4_ab_$T.log | sort | uniq -c; cat $O/r4_tests_$T.log
