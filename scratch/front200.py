import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi
from qtos_amd.config import PlannerConfig, REFERENCE_WALK_UNNORMALISED, scaled_phases
def show(tag, cfg):
    try:
        d, act = capi.analyze(cfg)
        print(tag, {k: getattr(d, k) for k, _ in d._fields_}, "max active", act.max())
    except Exception as e:
        print(tag, "ERR", e)
show("knots100", PlannerConfig.knots100())
show("5s dt .025", PlannerConfig.knots100(dt_base=0.025, dt_dynamic=0.025))
show("10s scaled", PlannerConfig.knots100(duration=10.0))
# two gait cycles: the walk table continued (the final stance of the first pass merges with the first stance of the second)
tab = []
for f in REFERENCE_WALK_UNNORMALISED:
    f = list(f); tab.append(f[:-1] + [f[-1] + f[0]] + f[1:])
show("10s two cycles", PlannerConfig.knots100(duration=10.0, phase_durations=scaled_phases(tab, 10.0)))
show("10s two cycles dt_rom .1", PlannerConfig.knots100(duration=10.0, dt_range_of_motion=0.1, phase_durations=scaled_phases(tab, 10.0)))
