#!/bin/bash
# SQ counters of the KKT kernel selected by QTOS_KKT=$1 (tag $2)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export QTOS_KKT=$1; T=$2
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_sq_$T.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_sq2_$T.log 2>&1
cd $R; python3 - <<PY
import csv, collections
for sub in ("pmc_sq_$T", "pmc_sq2_$T"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$O/" + sub + "/runc_counter_collection.csv")):
        if "k_kkt" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
