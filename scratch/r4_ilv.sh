#!/bin/bash
# two targets of a thread side by side in k_kkt3 MODE 1's phase-AB assembly: old library against new
R=$GRAFT_REPO_ROOT; cd $R
LIBS=libqtos_planner_old.so,libqtos_planner.so
AB_WLS=trot AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/'
AB_CFG=reference_compat AB_WLS=walk AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/' | sed 's/^/compat /'
QTOS_KKT=4 AB_WLS=walk AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/' | sed 's/^/KKT=4 /'
AB_WLS=walk AB_VAR=QTOS_LIB AB_VALS=$LIBS timeout 900 python scratch/ab5.py 2>&1 | grep -v amdgpu.ids | sed 's/; conv.*sha/ sha/' | sed 's/^/default /'
