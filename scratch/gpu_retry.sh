#!/bin/bash
# usage: scratch/gpu_retry.sh <timeout-seconds> '<command>' : gpurun with retries while the pod's GPU slots are busy (exit code 3)
T=$1; shift
for i in $(seq 1 30); do
  gpurun --timeout $T -- "$@" > /tmp/gpurun_last.txt 2>&1; rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.txt; then sleep 60; continue; fi
  break
done
tail -40 /tmp/gpurun_last.txt
