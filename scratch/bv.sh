#!/bin/bash
# dev build of a named variant: scratch/bv.sh name [flags] -> csrc/libqtos_planner_<name>.so (knots100 / compat fronts only)
cd /root/repo/quadruped-trajectory-optimization-stack_amd/csrc
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQTOS_DEV_SMALL "$@" qtos_planner.hip -o libqtos_planner_$n.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs Spill|ScratchSize|SGPRs Spill|  VGPRs:" | paste - - - - - | grep -E "error|kkt2ILi128ELb0" | sed 's/\[-Rpass[^]]*\]//g; s/.\/kkt2.hpp:[0-9]*:1: remark: //g'
