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
                # the schedule of the configured plan stretched to the new horizon (a custom phase table,
                # e.g. knots200's two-cycle one, keeps its shape)
                kw = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}
                kw["phase_durations"] = [[d * duration / cfg.duration for d in foot] for foot in cfg.phase_durations]
                kw["duration"] = duration
                cfg = PlannerConfig(**kw)
            p = capi.Planner(cfg, self.max_batch, self.device)
            p.set_kernel_events(False)   # (nobody reads per-kernel times behind this boundary: the event packets between the kernels cost a batch 1.9 %)
            self._push_terrain(p)
            self._planners[key] = p
        return self._planners[key]

    # ---- the boundary ----
    def solve_batch(self, args_list, map_id=None, warm=None, sample=True):
        """List of reference-style args dicts -> list of exit statuses.  Results in ``self.last``.

        Every ``./main`` call of the reference carries its own ``-duration``: the batch is grouped by
        horizon and every group is solved on the planner built for it (rows / nodes of ``self.last`` are
        lists when the horizons differ, arrays otherwise).  ``-r`` (scripts/main.py:119,203;
        QTOS/generateHeightField.py:373) is accepted and recorded in ``self.last["r"]``: see flags.py."""
        if not args_list:
            return []
        n = len(args_list)
        durs = [float(a.get('-duration') or self.cfg.duration) for a in args_list]
        order = {}
        for i, dv in enumerate(durs):
            order.setdefault(round(dv, 9), []).append(i)
        if warm is not None and len(order) > 1 and not isinstance(warm, (list, tuple)):
            # a warm row has the length of ITS horizon's variable vector: one array cannot serve two horizons
            raise ValueError("solve_batch: `warm` with mixed -duration values must be a list with one node vector per problem")
        statuses = [None] * n
        nodes_o, rows_o, iters_o, viol_o, t0_o = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
        for key, idx in order.items():
            P = self.planner(key)
            starts, goals, t0s = [], [], []
            for i in idx:
                s, g, t0 = flags.problem_arrays(args_list[i])
                starts.append(s)
                goals.append(g)
                t0s.append(t0)
            for c in range(0, len(idx), self.max_batch):
                sl = slice(c, c + self.max_batch)
                ii = idx[sl]
                nodes, status, iters, viol = P.plan(
                    np.array(starts[sl]), np.array(goals[sl]),
                    None if map_id is None else np.asarray(map_id)[ii],
                    None if warm is None else np.stack([np.asarray(warm[i], float) for i in ii]))
                rows = P.sample(nodes, np.array(t0s[sl]), self.cfg.hz) if sample else None
                for j, i in enumerate(ii):
                    statuses[i] = int(status[j])
                    nodes_o[i], iters_o[i], viol_o[i], t0_o[i] = nodes[j], iters[j], viol[j], t0s[c + j]
                    rows_o[i] = None if rows is None else rows[j]
        same = len(order) == 1
        # (one horizon solved in one call of the library -- the usual case: the sampler's array is the result; re-stacking the
        #  per-problem views copied 1.5 MB per plan twice over)
        whole = same and n <= self.max_batch
        self.last = dict(nodes=np.stack(nodes_o) if same else nodes_o,
                         rows=(rows if whole else (np.stack(rows_o) if same else rows_o)) if sample else None,
                         status=np.array(statuses), iters=np.array(iters_o), viol=np.array(viol_o), t0=np.array(t0_o),
                         r=[a.get('-r') for a in args_list])
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
