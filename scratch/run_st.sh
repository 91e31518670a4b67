#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for L in "$@"; do echo $L; QTOS_LIB=$L timeout 300 python scratch/stamps2.py 2>&1 | tail -6 | cut -c1-400; done > $O/r3_st.log 2>&1; cat $O/r3_st.log
