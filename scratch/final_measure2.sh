#!/bin/bash
# round-2 measurement pass: bench lines, rocprofv3 kernel stats, PMC passes (HBM traffic, SQ counters) -- each its own run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
mkdir -p $O; cd $R
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_${T}_$name.json 2> $O/bench_${T}_$name.err; tail -c 600 $O/bench_${T}_$name.json | head -c 0; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${T}_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("roofline",{}).get("avg_launch_ms"), d["config"].get("converged"), "/", d["config"].get("plans_timed"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
b compat --cpu-sample 0 --no-parity --transcription reference_compat
b exp5 --cpu-sample 0 --no-parity --workload exp5_step
b mixed --cpu-sample 0 --no-parity --workload mixed
b trot --cpu-sample 0 --no-parity --gait trot
b tol1e-3 --cpu-sample 0 --no-parity --tol 1e-3
b batch512 --cpu-sample 0 --no-parity --batch 512
b batch1024 --cpu-sample 0 --no-parity --batch 1024
b knots200 --cpu-sample 0 --no-parity --transcription knots200
b mpc --cpu-sample 0 --no-parity --transcription knots200 --workload mpc_random --steps 200
b mpc_1set --cpu-sample 0 --no-parity --transcription knots200 --workload mpc_random --steps 200 --inflight 1
b table --cpu-sample 0 --no-parity --init table
b inflight2 --cpu-sample 0 --no-parity --inflight 2
b nochord --cpu-sample 0 --no-parity --chord-tol 0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --cpu-sample 0 --no-parity > $O/bench_${T}_torchrun1.json 2>/dev/null; cut -c1-160 $O/bench_${T}_torchrun1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\(MFMA\|BUSY\|WAVE_CYCLES\|LDS_BANK\|LDS_IDX\|WAIT_INST\|WAIT_ANY\)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/counters_$T.txt; cat $O/counters_$T.txt; echo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o runc -- python3 $R/bench.py --cpu-sample 0 --no-parity > $O/prof_$T.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity > $O/pmc_fetch_$T.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity > $O/pmc_write_$T.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity > $O/pmc_sq_$T.log 2>&1
tail -2 $O/pmc_sq_$T.log
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity > $O/pmc_sq2_$T.log 2>&1
tail -2 $O/pmc_sq2_$T.log
find $O/prof_$T $O/pmc_fetch_$T $O/pmc_write_$T $O/pmc_sq_$T $O/pmc_sq2_$T -name "*.csv" | head -20
