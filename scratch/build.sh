#!/bin/bash
# diagnostic build: scratch/build.sh [extra flags] -> csrc/libqtos_planner_stamps.so (s_memtime stamps per phase and wave,
# fronts up to 128 slots only; read with scratch/stamps2.py).  The product library is built by csrc/Makefile.
cd "$(dirname "$0")/../quadruped-trajectory-optimization-stack_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQTOS_DEV_SMALL -DQTOS_STAMPS "$@" qtos_planner.hip -o ${OUT:-libqtos_planner_stamps.so} 2>&1 | grep -E "error"
ls -la ${OUT:-libqtos_planner_stamps.so} | awk '{print $5, $9}'
