#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
QTOS_DEBUG_SYMBOLIC=1 QTOS_KRON=1 python -c "
import sys; sys.path.insert(0,'.')
from qtos_amd import capi
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.knots100(), max_batch=8)
" 2>&1 | grep -i kron
bash scratch/r4_kron.sh
