"""LocalPlanner: the Python face of the drop-in boundary.

In the reference one plan is ``subprocess.run("docker exec <id> ./main " + cmd_args(args))``
followed by ``docker cp <id>:.../build/traj.csv ./data/traj/towr.csv`` (scripts/main.py:49-57,
90-92).  ``LocalPlanner.solve(args)`` takes the same ``args`` dict, solves on the GPU through the
C ABI (capi.py -> csrc/libqtos_planner.so) and writes the same 37-column CSV; the return value is
the process exit status the reference checks (0 = solved, scripts/main.py:93-103).
``solve_batch`` is the batched form (the reference's own batch is 32 concurrent ``docker exec``
calls, QTOS/generateHeightField.py:344-386).
"""
import os

import numpy as np

from . import capi, csvio, flags, heightfield
from .config import PlannerConfig

TOWR_HEIGHTFIELD = "./data/heightfields/from_pybullet/towr_heightfield.txt"  # QTOS/utils.py:21-22
TRAJ_OUT = "./data/traj/towr.csv"                                             # QTOS/utils.py:16


class LocalPlanner:
    def __init__(self, cfg=None, max_batch=256, device=0):
        self.cfg = cfg or PlannerConfig.reference_compat()
        self.max_batch, self.device = max_batch, device
        self._planners = {}
        self._terrain = None  # (maps[n][nx][ny], cell, x0, y0)
        self.last = None

    # ---- terrain ----
    def set_heightfield(self, height_xy, cell, x0=-1.0, y0=-1.0):
        """height_xy[ix][iy] (or a stack of such maps) in the solver file's orientation."""
        self._terrain = None if height_xy is None else (np.asarray(height_xy, float), cell, x0, y0)
        for p in self._planners.values():
            self._push_terrain(p)

    def load_heightfield_file(self, path=TOWR_HEIGHTFIELD, resolution=None):
        """Read the file the reference pushes into the container (scripts/main.py:77-78)."""
        arr = heightfield.read_height_file(path)
        cell = resolution if resolution else 2.0 / arr.shape[1]
        self.set_heightfield(arr, cell)

    def _push_terrain(self, p):
        if self._terrain is None:
            p.set_heightfields(None, 1.0)
        else:
            maps, cell, x0, y0 = self._terrain
            p.set_heightfields(maps, cell, x0, y0)

    # ---- planners are cached per plan duration (-duration flag, scripts/main.py:119-120) ----
    def planner(self, duration=None):
        duration = float(duration or self.cfg.duration)
        key = round(duration, 9)
        if key not in self._planners:
            cfg = self.cfg
            if abs(duration - cfg.duration) > 1e-12:
                kw = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__ if k != "phase_durations"}
                kw["duration"] = duration
                cfg = PlannerConfig(**kw)
            p = capi.Planner(cfg, self.max_batch, self.device)
            self._push_terrain(p)
            self._planners[key] = p
        return self._planners[key]

    # ---- the boundary ----
    def solve_batch(self, args_list, map_id=None, warm=None, sample=True):
        """List of reference-style args dicts -> list of exit statuses.  Results in ``self.last``."""
        if not args_list:
            return []
        dur = args_list[0].get('-duration') or None
        P = self.planner(dur)
        starts, goals, t0s = [], [], []
        for a in args_list:
            s, g, t0 = flags.problem_arrays(a)
            starts.append(s)
            goals.append(g)
            t0s.append(t0)
        statuses, nodes_all, iters_all, viol_all = [], [], [], []
        for i in range(0, len(args_list), self.max_batch):
            sl = slice(i, i + self.max_batch)
            nodes, status, iters, viol = P.plan(
                np.array(starts[sl]), np.array(goals[sl]),
                None if map_id is None else np.asarray(map_id)[sl],
                None if warm is None else np.asarray(warm)[sl])
            statuses += [int(s) for s in status]
            nodes_all.append(nodes)
            iters_all.append(iters)
            viol_all.append(viol)
        nodes = np.concatenate(nodes_all)
        rows = None
        if sample:
            rows = np.concatenate([P.sample(nodes[i:i + self.max_batch], np.array(t0s[i:i + self.max_batch]), self.cfg.hz)
                                   for i in range(0, len(args_list), self.max_batch)])
        self.last = dict(nodes=nodes, rows=rows, status=np.array(statuses), iters=np.concatenate(iters_all),
                         viol=np.concatenate(viol_all), t0=np.array(t0s))
        return statuses

    def solve(self, args, out_csv=TRAJ_OUT):
        """One plan; writes the CSV where the reference's ``docker cp`` would put it."""
        status = self.solve_batch([args])[0]
        if out_csv:
            d = os.path.dirname(out_csv)
            if d:
                os.makedirs(d, exist_ok=True)
            csvio.write_csv(out_csv, self.last["rows"][0])
        return status

    def close(self):
        for p in self._planners.values():
            p.close()
        self._planners = {}
