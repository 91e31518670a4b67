#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for L in "$@"; do echo "== $L"; QTOS_LIB=$L timeout 300 python scratch/stamps2.py 2>&1 | grep -E "^wave  0|^wave  1:|^wave 13|^wave 15|stage total" | cut -c1-100; done > $O/r4_abl.log 2>&1; cat $O/r4_abl.log
