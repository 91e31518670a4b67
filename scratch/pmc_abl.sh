#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in base NOASM NOEXTRACT NORHS; do
  export QTOS_LIB=libqtos_planner_$v.so
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $O/pmc_abl_$v -o runc -- python3 $R/scratch/one_plan.py > $O/pmc_abl_$v.log 2>&1
  tail -1 $O/pmc_abl_$v.log
done
