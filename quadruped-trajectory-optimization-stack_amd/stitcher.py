"""Receding-horizon stitching of plan CSVs: counterpart of QTOS/combiner.py (``_state``,
``combine``, ``_truncate_csv``) and QTOS/utils.py:495-541 (``look_ahead``, ``zero_filter``),
operating on in-memory row arrays (rows x 37) instead of files.

``mode="reference"`` reproduces the reference bug-for-bug: ``pd.read_csv`` without ``header=None``
eats the first row of both the old and the new CSV (QTOS/combiner.py:131,305), so the stitched
plan keeps the OLD plan's hand-over row and starts the new plan at its second row (visible in
data/traj/towr.csv rows 1254/1255).  ``mode="clean"`` keeps every row.
"""
import numpy as np

EE_KEYS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")


def look_ahead_index(rows, start_time=0.0, timesteps=6000, decimal_roundoff=3):
    """Index of the row the reference's reader yields next, and its ``stop_idx``
    (QTOS/utils.py:495-521): first row with start_time <= round(t, 3), then ``timesteps`` ahead."""
    t = np.round(np.asarray(rows)[:, 0], decimal_roundoff)
    hit = np.nonzero(start_time <= t)[0]
    if len(hit) == 0:
        raise StopIteration("start_time beyond the plan")
    i = int(hit[0])
    return i + timesteps, i + 1


def zero_filter(values, tol=1e-4):
    v = np.array(values, dtype=float)
    v[np.abs(v) < tol] = 0.0
    return v.tolist()


def row_state(row):
    """Column contract of a plan row (QTOS/combiner.py:263-274)."""
    r = np.asarray(row, float)[1:]
    return {"CoM": r[0:3].tolist(), "orientation": r[3:6].tolist(), "FL_FOOT": r[6:9].tolist(),
            "FR_FOOT": r[9:12].tolist(), "HL_FOOT": r[12:15].tolist(), "HR_FOOT": r[15:18].tolist(),
            "CoM_vel": r[18:21].tolist(), "CoM_vel_ang": r[21:24].tolist()}


class Stitcher:
    def __init__(self, lookahead=2750, hz=1000, height_set=(0.0,), mode="reference"):
        self.lookahead_original = lookahead
        self.lookahead = lookahead
        self.hz = hz
        self.height_set = set(float(h) for h in height_set)
        self.mode = mode
        self.cutoff_idx = 0
        self.next_traj_step = 0

    def legs_in_contact(self, state, tol=6):
        return all(round(state[k][2], tol) in self.height_set for k in EE_KEYS)

    def state(self, rows, last_timestep):
        """Hand-over state: the row ``lookahead`` steps ahead of ``last_timestep``, advanced until every
        foot stands on a known terrain height (QTOS/combiner.py:245-296); falls back to the
        un-shifted row when the plan ends first."""
        rows = np.asarray(rows)
        self.lookahead = self.lookahead_original
        idx, step = look_ahead_index(rows, last_timestep, self.lookahead)
        state = None
        while True:
            if idx >= len(rows):
                self.lookahead = self.lookahead_original
                idx, step = look_ahead_index(rows, last_timestep, self.lookahead)
                state = row_state(rows[idx])
                break
            state = row_state(rows[idx])
            if self.legs_in_contact(state):
                break
            self.lookahead += 1
            idx += 1
        state = {k: zero_filter(v) for k, v in state.items()}
        self.next_traj_step = step + self.lookahead - 1
        return state

    def plan_args(self, args, state, runtime, goal):
        """Fill the solver flags like Combiner.plan (QTOS/combiner.py:166-179)."""
        args = dict(args)
        args['-s'], args['-s_ang'] = state["CoM"], state["orientation"]
        args['-e1'], args['-e2'] = state["FL_FOOT"], state["FR_FOOT"]
        args['-e3'], args['-e4'] = state["HL_FOOT"], state["HR_FOOT"]
        args['-t'] = runtime + self.lookahead / self.hz
        args['-g'] = list(goal)
        args['s_vel'], args['s_ang_vel'] = state["CoM_vel"], state["CoM_vel_ang"]
        return args

    def combine(self, old_rows, new_rows):
        """old[cutoff-1 : next_traj_step] ++ new  (QTOS/combiner.py:125-135, 298-312)."""
        old_rows, new_rows = np.asarray(old_rows), np.asarray(new_rows)
        if self.cutoff_idx <= 0:
            start = self.cutoff_idx = 0
        else:
            start = self.cutoff_idx - 1
        end = self.next_traj_step
        if self.mode == "reference":
            old = old_rows[1:][start:end]   # read_csv consumed row 0 as a header
            new = new_rows[1:]
        else:
            old, new = old_rows[start:end], new_rows
        return np.concatenate([old, new], axis=0)
