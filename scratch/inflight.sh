#!/bin/bash
for w in exp1_flat mixed; do
  for f in 1 2 3; do
    python bench.py --steps 12 --warmup 2 --cpu-sample 0 --workload $w --inflight $f 2>&1 | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('$w', $f, d['value'], d['ms_per_step'], d['config']['converged'])" || tail -3 /tmp/o.json
  done
done
