#!/bin/bash
# LDS counters of k_kkt2 with the old assembly and with the Kronecker assembly (QTOS_KRON=0 / 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
export QTOS_KRON=$v; T=kron$v
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$T -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-trot > $O/pmc_$T.log 2>&1
done
cd $R; python3 - <<PY
import csv, collections
for sub in ("pmc_kron0", "pmc_kron1"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$O/" + sub + "/runc_counter_collection.csv")):
        if "k_kkt" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(sub, k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
