#!/bin/bash
# k_step A/B: scratch/ab_step.sh <lib> ... — kernel-trace stats of a walk bench run per development library (QTOS_LIB)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  export QTOS_LIB=$L QTOS_KKT=2
  rm -rf $O/abp_$L
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abp_$L -o p -- python3 $R/bench.py --steps 20 --cpu-sample 0 --no-trot --no-parity > $O/abp_$L.log 2>&1
  echo "== $L"; tail -1 $O/abp_$L.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('converged'))"
  python3 - <<PY
import csv,glob
f=glob.glob('$O/abp_$L/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print('%-60s calls %6s avg %10.1f us total %5.1f %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
  rm -f $O/abp_$L/*kernel_trace.csv
done
