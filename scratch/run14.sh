#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3; do python bench.py --cpu-sample 0 --no-parity | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['converged'], d['config']['iterations_mean'], d['roofline']['avg_launch_ms'])"; done
python bench.py --cpu-sample 0 --no-parity --workload mixed | cut -c1-200
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --cpu-sample 0 --no-parity 2>/dev/null | cut -c1-220
