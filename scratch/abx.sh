#!/bin/bash
# usage: abx.sh tag lib1 lib2 ... : quick parity tests of every lib but the first + A/B timing of the listed libs (no full suite)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
first=1
for L in "$@"; do
  if [ $first = 1 ]; then first=0; continue; fi
  echo $L; QTOS_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kkt_solve or knots100_batch or golden_inputs or chord or step_terrain or two_phase" 2>&1 | tail -3
done
timeout 600 python scratch/ab2.py "$@" > $O/r3_ab_$T.log 2>&1; grep kkt $O/r3_ab_$T.log | sort
