#!/bin/bash
# SQ counters of the KKT kernel of a development library: scratch/r5_pmc_lib.sh <lib> <kkt> [gait]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export QTOS_LIB=libqtos_$1.so QTOS_KKT=$2; T=$1_$2; G=${3:-walk}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_sq_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot --gait $G > $O/pmc_sq_$T.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot --gait $G > $O/pmc_sq2_$T.log 2>&1
cd $R; python3 - <<PY
import csv, collections
for sub in ("pmc_sq_$T", "pmc_sq2_$T"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        rows = list(csv.DictReader(open("$O/" + sub + "/runc_counter_collection.csv")))
    except Exception as e:
        print(sub, "failed", e); continue
    for r in rows:
        if "k_kkt" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        d = {c: sum(x) / len(x) for c, x in v.items()}
        print(k, {c: round(x) for c, x in d.items()}, "n", len(next(iter(v.values()))))
        if "SQ_WAVE_CYCLES" in d:
            waves = 12 * 256 if "kkt5" in k else 16 * 256
            cyc = 4 * d["SQ_WAVE_CYCLES"] / waves
            print("   cycles/launch %.0f  mfma busy %.3f  wait_any %.3f  lds conflict %.3f  mfma MFLOP/problem %.1f" % (cyc, d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), d["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / 256 / 1e6))
        if "SQ_INSTS_VALU" in d:
            print("   per problem: VALU %.0f  LDS %.0f  SALU %.0f  SMEM %.0f ; valu active quad-cycles %.0f" % (d["SQ_INSTS_VALU"] / 256, d["SQ_INSTS_LDS"] / 256, d.get("SQ_INSTS_SALU", 0) / 256, d.get("SQ_INSTS_SMEM", 0) / 256, d["SQ_ACTIVE_INST_VALU"] / 256))
PY
