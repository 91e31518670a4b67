#!/bin/bash
# usage: run4.sh tag lib1 lib2 ... : gpu tests with the default lib, then A/B timing of the listed libs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
timeout 600 python -m pytest tests -m gpu -x -q --timeout 200 > $O/r2_pytest_$T.log 2>&1; tail -4 $O/r2_pytest_$T.log
for L in "$@"; do QTOS_LIB=$L timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch" 2>&1 | tail -1; done
timeout 600 python scratch/ab.py "$@" > $O/r2_ab_$T.log 2>&1; tail -$(( 2 * $# )) $O/r2_ab_$T.log
