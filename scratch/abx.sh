#!/bin/bash
# usage: abx.sh tag lib1 lib2 ... : quick parity tests + A/B timing of the listed libs (no full suite)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
[ -x scratch/ub/t_perm ] && scratch/ub/t_perm | head -2
for L in "$@"; do echo $L; QTOS_LIB=$L timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord" 2>&1 | tail -3; done
timeout 600 python scratch/ab2.py "$@" > $O/r2_ab_$T.log 2>&1; grep kkt $O/r2_ab_$T.log | sort
