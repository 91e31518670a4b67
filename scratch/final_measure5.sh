#!/bin/bash
# round-5 measurement pass: bench lines, rocprofv3 kernel stats, PMC passes (HBM traffic, SQ counters) -- each its own run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
mkdir -p $O; cd $R
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_${T}_$name.json 2> $O/bench_${T}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${T}_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("roofline",{}).get("avg_launch_ms"), d["config"].get("converged"), "/", d["config"].get("plans_timed"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
X="--cpu-sample 0 --no-parity --no-trot"
b full_system $X --full-system
b compat $X --transcription reference_compat
b exp5 $X --workload exp5_step
b mixed $X --workload mixed
b exp5_lanes3 $X --workload exp5_step --inflight 3
b mixed_lanes3 $X --workload mixed --inflight 3
b trot $X --gait trot
b tol1e-3 $X --tol 1e-3
b batch512 $X --batch 512
b batch1024 $X --batch 1024
b exp5_batch1024 $X --workload exp5_step --batch 1024
b mixed_batch1024 $X --workload mixed --batch 1024
b knots200 $X --transcription knots200
b mpc $X --transcription knots200 --workload mpc_random --steps 200
b mpc_1set $X --transcription knots200 --workload mpc_random --steps 200 --inflight 1
b table $X --init table
b lanes2 $X --inflight 2
b nochord $X --chord-tol 0
b steps500 $X --steps 500
b torchrun1 $X --force-torchrun
# k_kkt5 (QTOS_KKT=6: two stages per set of barriers) next to the defaults, same box
QTOS_KKT=6 b kkt5_walk $X
QTOS_KKT=6 b kkt5_trot $X --gait trot
QTOS_KKT=6 b kkt5_compat $X --transcription reference_compat
QTOS_KKT=6 b kkt5_knots200 $X --transcription knots200
QTOS_KKT=2 b kkt2_trot $X --gait trot
QTOS_KKT=2 b kkt2_walk $X
# the system of round 4 (swing mid nodes as unknowns) on the same box
b no_swing $X --full-swings
b no_swing_trot $X --full-swings --gait trot
# ... and without the short stages below 128 slots: the configuration of round 4's final pass
QTOS_SHORT_STAGES=0 b r4_system $X --full-swings --plain-mu
QTOS_SHORT_STAGES=0 b r4_system_trot $X --full-swings --plain-mu --gait trot
# the barrier parameter's update of rounds 1 - 4 alone
b plain_mu $X --plain-mu
b plain_mu_trot $X --plain-mu --gait trot
b plain_mu_exp5 $X --plain-mu --workload exp5_step
b plain_mu_mixed $X --plain-mu --workload mixed
b plain_mu_knots200 $X --plain-mu --transcription knots200
b superlinear_mu_mpc $X --superlinear-mu --transcription knots200 --workload mpc_random --steps 200
QTOS_SHORT_STAGES=0 b no_short $X
QTOS_SHORT_STAGES=0 b no_short_trot $X --gait trot
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o runc -- python3 $R/bench.py --cpu-sample 0 --no-parity --no-trot > $O/prof_$T.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_fetch_$T.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_write_$T.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_sq_$T.log 2>&1
tail -2 $O/pmc_sq_$T.log
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_sq2_$T.log 2>&1
tail -2 $O/pmc_sq2_$T.log
find $O/prof_$T $O/pmc_fetch_$T $O/pmc_write_$T $O/pmc_sq_$T $O/pmc_sq2_$T -name "*.csv" | head -20
# the trot (the gait the metric names): kernel stats and counters of its own
bash $R/scratch/r5_trot_prof.sh ${T}trot > /dev/null 2>&1
# k_kkt5: SQ counters
export QTOS_KKT=6
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}k5 -o runc -- python3 $R/bench.py --cpu-sample 0 --no-parity --no-trot > $O/prof_${T}k5.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq_${T}k5 -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_sq_${T}k5.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_${T}k5 -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_sq2_${T}k5.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${T}k5 -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_fetch_${T}k5.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${T}k5 -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_write_${T}k5.log 2>&1
unset QTOS_KKT
ls $O | grep ${T} | head -80
