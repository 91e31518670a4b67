#!/bin/bash
# round-6 measurement pass on ONE box: bench lines of every configuration, then per gait rocprofv3 kernel stats + PMC passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
mkdir -p $O; cd $R
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_${T}_$name.json 2> $O/bench_${T}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${T}_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("roofline",{}).get("avg_launch_ms"), d.get("gap_ms_per_step"), d["config"].get("converged"), "/", d["config"].get("plans_timed"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
b driver_cmd --steps 20 --warmup 5
X="--cpu-sample 0 --no-parity --no-second-gait"
b trot $X
b walk $X --gait walk
b trot_no_pattern $X --no-pattern
b walk_no_pattern $X --gait walk --no-pattern
b full_system $X --gait walk --full-system
b compat $X --gait walk --transcription reference_compat
b compat_trot $X --transcription reference_compat
b exp5 $X --workload exp5_step
b mixed $X --workload mixed
b exp5_no_pattern $X --workload exp5_step --no-pattern
b mixed_no_pattern $X --workload mixed --no-pattern
b exp5_lanes3 $X --workload exp5_step --inflight 3
b mixed_lanes3 $X --workload mixed --inflight 3
b tol1e-3 $X --tol 1e-3
b batch512 $X --batch 512
b batch1024 $X --batch 1024
b exp5_batch1024 $X --workload exp5_step --batch 1024
b mixed_batch1024 $X --workload mixed --batch 1024
b knots200 $X --transcription knots200
b mpc $X --transcription knots200 --workload mpc_random --steps 200
b mpc_no_pattern $X --transcription knots200 --workload mpc_random --steps 200 --no-pattern
b mpc_1set $X --transcription knots200 --workload mpc_random --steps 200 --inflight 1
b superlinear_mu_mpc $X --superlinear-mu --transcription knots200 --workload mpc_random --steps 200
b table $X --init table
b lanes2 $X --inflight 2
b nochord $X --chord-tol 0
b steps500 $X --steps 500
b torchrun1 $X --force-torchrun
QTOS_ORDER=0 b order0_walk $X --gait walk
QTOS_ORDER=0 b order0_trot $X
QTOS_ORDER=0 b order0_compat $X --gait walk --transcription reference_compat
QTOS_ORDER=0 b order0_compat_trot $X --transcription reference_compat
QTOS_ORDER=0 b order0_exp5 $X --workload exp5_step
QTOS_ORDER=0 b order0_mixed $X --workload mixed
QTOS_ORDER=0 b order0_knots200 $X --transcription knots200
QTOS_ORDER=0 b order0_mpc $X --transcription knots200 --workload mpc_random --steps 200
QTOS_KKT=6 b kkt5_walk $X --gait walk
QTOS_KKT=6 b kkt5_trot $X
QTOS_KKT=2 b kkt2_trot $X
QTOS_KKT=2 b kkt2_walk $X --gait walk
# the default command itself (both legs: the trot's k_kkt3<96, 1> and the walk's k_kkt3<112, 1> are separate rows) under rocprofv3
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}default -o runc -- python3 $R/bench.py --cpu-sample 0 --no-parity > $O/prof_${T}default.log 2>&1)
bash $R/scratch/r6_gait_prof.sh ${T}trot trot > /dev/null 2>&1
bash $R/scratch/r6_gait_prof.sh ${T}walk walk > /dev/null 2>&1
ls $O | grep ${T} | wc -l
