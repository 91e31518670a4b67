#!/bin/bash
# bench lines (walk, trot, knots200, reference_compat) under the environment given on the command line: scratch/ab_env.sh VAR=val ...
for G in "--gait walk" "--gait trot" "--transcription knots200" "--transcription reference_compat" "--workload exp5_step" "--workload mixed"; do
  env "$@" timeout 300 python bench.py --steps 30 --cpu-sample 0 --no-parity --no-trot $G 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-34s %-16s kkt %.4f ms chord %.4f  plans/s %8.0f  ms/step %.3f conv %s/%s stages %s front %s' % ('$G', r['kernel'], r['avg_launch_ms'], r.get('chord_avg_launch_ms') or 0, d['value'], d['ms_per_step'], d['config']['converged'], d['config']['plans_timed'], d['config']['kkt_stages'], d['config']['front']))"
done
