#!/bin/bash
# dev build: scratch/build.sh [extra flags] -> libqtos_planner.so (+ resource usage of k_kkt2<128>), and the stamps lib in parallel
cd /root/repo/quadruped-trajectory-optimization-stack_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQTOS_DEV_SMALL -DQTOS_STAMPS "$@" qtos_planner.hip -o libqtos_planner_stamps.so 2>&1 | grep -E "error" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQTOS_DEV_SMALL "$@" qtos_planner.hip -o libqtos_planner.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs Spill|ScratchSize|SGPRs Spill" | paste - - - - | grep -E "error|kkt2ILi128ELb0|kkt2ILi112ELb0" | sed 's/\[-Rpass[^]]*\]//g; s/.\/kkt2.hpp:[0-9]*:1: remark: //g'
wait
ls -la *.so | awk '{print $6,$7,$8,$9}'
