#!/bin/bash
# round 6: the driver's exact bench command on N fresh leases (one gpurun call each), lines into profiles/r06_driver_cmd_<i>.json
# usage: scratch/r6_driver_cmd.sh <first> <last> [extra bench flags -> suffix "_x"]
A=$1; B=$2; shift 2
for i in $(seq $A $B); do
  scratch/gpu_retry.sh 900 "python3 bench.py --gpus 1 --steps 20 --warmup 5 $* > gpurun_out/r06_driver_cmd_$i.json 2> gpurun_out/r06_driver_cmd_$i.err" > /tmp/r6_drv_$i.log 2>&1
  python3 -c "import json;json.load(open('gpurun_out/r06_driver_cmd_$i.json'))" && cp gpurun_out/r06_driver_cmd_$i.json profiles/r06_driver_cmd_$i.json
done
