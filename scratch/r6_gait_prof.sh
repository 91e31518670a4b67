#!/bin/bash
# round 6: kernel evidence of ONE gait's bench command (the line's main leg only): rocprofv3 kernel stats + the separate PMC passes
# usage: scratch/r6_gait_prof.sh <tag> <walk|trot> [extra bench flags]      (collect: scratch/collect_kernel_evidence.py <tag> <prefix>)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; G=$2; shift 2
mkdir -p $O
F="--gait $G --cpu-sample 0 --no-parity --no-second-gait $*"
cd $R
timeout 600 python3 bench.py $F --steps 100 > $O/bench_${T}.json 2> $O/bench_${T}.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o runc -- python3 $R/bench.py $F > $O/prof_$T.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 $F > $O/pmc_fetch_$T.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 $F > $O/pmc_write_$T.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 $F > $O/pmc_sq_$T.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 $F > $O/pmc_sq2_$T.log 2>&1
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
