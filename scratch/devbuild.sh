#!/bin/bash
# development build: scratch/devbuild.sh <name> [extra flags] -> csrc/libqtos_<name>.so with the benchmark's fronts only (112, 128
# slots: a quarter of the compile time); select it with QTOS_LIB=libqtos_<name>.so.  The product library is built by csrc/Makefile.
cd "$(dirname "$0")/../quadruped-trajectory-optimization-stack_amd/csrc"
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQTOS_DEV_F128 "$@" qtos_planner.hip -o libqtos_$N.so 2>&1 | grep -E "error" 
ls -la libqtos_$N.so | awk '{print $5, $9}'
