import sys; sys.path.insert(0, '.')
import os
which = sys.argv[1]
if which == "torch_first":
    import torch; print("torch first:", torch.cuda.is_available(), torch.cuda.device_count())
from qtos_amd import capi
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.reference_compat(), max_batch=2)
import torch
print(which, "-> torch sees", torch.cuda.is_available(), torch.cuda.device_count())
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
