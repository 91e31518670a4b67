"""Receding-horizon re-planning: counterpart of the reference's planning thread.

``ReplanLoop`` mirrors ``scripts/main.py``: ``_run`` (26-103: first plan from the rest pose toward
``spine_step``, then the update thread) and ``_update`` (26-62), whose loop alternates three states

    plan     ``Combiner.plan``: hand-over state = the row ``lookahead`` steps ahead of the robot's clock in the
             plan being executed, advanced until every foot stands on a known terrain height
             (QTOS/combiner.py:158-179, 245-296); goal = ``Global_Planner.pop()`` (LIFO,
             QTOS/planner.py:232-239); one solver call
    wait     until the consumer has eaten ``f_steps`` rows of the current plan (scripts/main.py:52)
    stitch   ``Combiner.combine``: old[cutoff-1 : next_traj_step] ++ new (QTOS/combiner.py:125-135), after
             which the consumer restarts its row counter (scripts/run.py:177-183)

and stops when the start of the last plan is within 0.1 m of its goal (scripts/main.py:40-46).  The
solver call is ``LocalPlanner.solve_batch`` (one GPU solve) instead of ``docker exec ./main``; the
consumer is whoever calls ``tick(step, runtime)`` with the reference's two counters (``RUN.step``,
``ROBOT_CFG.runtime``) -- ``run()`` drives it with a consumer that tracks the plan perfectly.

``ShiftedWindows`` is the batched form used for BASELINE configs[4]: many independent robots (windows) on one
GPU, every step = one replan of every window with the hand-over semantics above.

Starting point of a replan.  ``qtos_shift_warm`` builds the time-shifted previous plan and both
classes can use it (``shifted_warm_start=True`` / ``warm="shifted"``), but it is NOT the default: the reference
restarts its gait schedule with every plan, so the previous plan read 2.5 s later lifts other feet at other times
than the new schedule, and measured on the randomized heightfields (scratch/shift_exp.py, 64 windows x 6 replans,
200-knot plans) the shifted plan needs 4.8-5.6 Newton iterations (slowest window 9-16) against 4.1-4.2 (slowest 5-6)
from towr's straight-line guess, whatever the slack push and whichever variable sets are shifted.  A batch waits
for its slowest window, so the loops start cold, like the reference's solver.
"""
import numpy as np

from . import flags
from .stitcher import Stitcher, row_state


class ReplanLoop:
    def __init__(self, local_planner, global_planner, args, lookahead=3750, f_steps=2500, hz=1000,
                 height_set=(0.0,), mode="reference", shifted_warm_start=False):
        self.lp, self.gp = local_planner, global_planner
        self.args = dict(args)
        self.f_steps, self.hz = int(f_steps), hz
        self.st = Stitcher(lookahead=lookahead, hz=hz, height_set=height_set, mode=mode)
        self.shifted = shifted_warm_start
        self.plan = None          # rows of the plan being executed (the reference's ./data/traj/towr.csv)
        self.new = None           # rows of the plan waiting to be stitched (/tmp/towr.csv)
        self.nodes = None         # nodes of the newest plan and the time stamp of its first row
        self.nodes_t0 = 0.0
        self._wait = False
        self.done = False
        self.goal_diff = np.inf
        self.events = []          # (event, runtime) log: "plan", "stitch", "done"
        self.statuses = []

    # scripts/main.py:81-92 + Combiner.plan_init (QTOS/combiner.py:137-156)
    def start(self):
        a = self.args
        a['-s'], a['-s_ang'] = [0, 0, 0.24], [0, 0, 0]
        a['-e1'], a['-e2'] = [0.21, 0.19, 0.0], [0.21, -0.19, 0.0]
        a['-e3'], a['-e4'] = [-0.21, 0.19, 0.0], [-0.21, -0.19, 0.0]
        a['-g'] = [float(v) for v in self.gp.spine_step(np.array(a['-s'], float), 0.0)]
        status = self.lp.solve_batch([a])[0]
        self.statuses.append(status)
        if status != 0:
            raise RuntimeError("first plan failed (the reference exits here: scripts/main.py:101-103)")
        self.plan = np.array(self.lp.last["rows"][0])
        self.nodes, self.nodes_t0 = np.array(self.lp.last["nodes"][0]), 0.0
        self.events.append(("plan", 0.0))
        return status

    # one pass of the loop body of _update (scripts/main.py:34-62) with the consumer's counters
    def tick(self, step, runtime):
        if self.done:
            return "done"
        self.st.cutoff_idx = int(step)
        last_t = round(runtime, 3)
        self.goal_diff = float(np.linalg.norm(np.array(self.args['-s'])[0:2] - np.array(self.args['-g'])[0:2]))
        self.gp.update(last_t)
        if self.gp.max_t < last_t - 5.0:            # QTOS/combiner.py:224-226
            self.done = True
        if self.goal_diff < 0.1 or self.done:
            self.done = True
            self.events.append(("done", runtime))
            return "done"
        if not self._wait:
            state = self.st.state(np.round(self.plan, 6), last_t)     # the CSV carries 6 digits
            if not self.gp.empty():
                _, goal = self.gp.pop()
                goal = [float(v) for v in goal]
            else:                                                       # Combiner._step (QTOS/combiner.py:229-238)
                pos = np.array(state["CoM"])
                d = np.clip(np.array(self.gp.robot_goal) - pos, -self.gp.step_size, self.gp.step_size)
                goal = [float(pos[0] + d[0]), float(pos[1] + d[1]), 0.24]
            self.args = self.st.plan_args(self.args, state, runtime, goal)
            warm = None
            if self.shifted and self.nodes is not None:
                s, g, t0 = flags.problem_arrays(self.args)
                off = max(self.args['-t'] - self.nodes_t0, 0.0)
                warm = self.lp.planner(self.args.get('-duration')).shift_warm(self.nodes[None], off, np.array(s)[None], np.array(g)[None])
            self.statuses.append(self.lp.solve_batch([self.args], warm=warm)[0])   # (the reference ignores this status: scripts/main.py:50)
            self.new = np.array(self.lp.last["rows"][0])
            self.nodes, self.nodes_t0 = np.array(self.lp.last["nodes"][0]), float(self.args['-t'])
            self._wait = True
            self.events.append(("plan", runtime))
            return "plan"
        if self.st.cutoff_idx >= self.f_steps:
            self.plan = self.st.combine(self.plan, self.new)
            self._wait = False
            self.events.append(("stitch", runtime))
            return "stitch"       # the consumer re-opens the plan: its row counter restarts at 0
        return None

    def run(self, max_plans=8, dt_tick=0.05):
        """Drive the loop with a consumer that follows the plan exactly: one row per millisecond."""
        self.start()
        step, runtime = 0, 0.0
        while not self.done and sum(1 for e in self.events if e[0] == "plan") < max_plans:
            ev = self.tick(step, runtime)
            if ev == "stitch":
                step = 0
            adv = int(round(dt_tick * self.hz))
            step += adv
            runtime += adv / self.hz
        return self.plan


class ShiftedWindows:
    """B independent receding windows on one planner: every ``replan()`` starts every window from the row
    ``advance`` seconds into its newest plan (= the reference's hand-over row: a new plan starts ``lookahead``
    rows ahead of the robot's clock and the next one is asked for ``f_steps`` rows later, i.e. ``f_steps`` rows
    into the newest plan), moved on until all four feet are in contact; goals move with the windows; the
    previous plan shifted by that time is the warm start.  Everything stays on the device (torch tensors)."""

    def __init__(self, planner, start, goal_step, map_id=None, advance=2.5, search=0.4, stream=None, warm="none", x_range=None):
        import torch
        self.torch = torch
        self.P = planner
        dev = torch.device("cuda", planner.device if hasattr(planner, "device") else 0)
        self.dev = dev
        B = len(start)
        self.B = B
        f64 = dict(dtype=torch.float64, device=dev)
        self.start = torch.as_tensor(np.asarray(start), **f64).contiguous()
        self.goal_step = torch.as_tensor(np.asarray(goal_step), **f64).contiguous()     # (B, 3): per-plan displacement
        self.goal = self.start[:, 0:3] + self.goal_step
        self.goal[:, 2] = 0.24
        self.goal = self.goal.contiguous()
        self.map_id = None if map_id is None else torch.as_tensor(np.asarray(map_id), dtype=torch.int32, device=dev).contiguous()
        self.nodes = torch.empty((B, planner.n), **f64)
        self.prev = torch.empty((B, planner.n), **f64)
        self.warm = torch.empty((B, planner.n), **f64)
        self.status = torch.empty((B,), dtype=torch.int32, device=dev)
        self.iters = torch.empty((B,), dtype=torch.int32, device=dev)
        self.viol = torch.empty((B,), **f64)
        self.advance, self.hz = float(advance), 1000.0
        self.n_search = int(round(search * self.hz))
        self.rows = torch.empty((B, int(round(advance * self.hz)) + self.n_search + 1, 37), **f64)
        self.t0 = torch.zeros((B,), **f64)
        self.offset = torch.zeros((B,), **f64)
        self.stream = stream if stream is not None else torch.cuda.current_stream(dev)
        self.have_plan = False
        self.x_range = x_range     # (lo, hi): a window that walks past an end of its heightfield turns round (no resets)
        self.warm_mode = warm      # "none": towr's straight-line guess (default); "shifted": the time-shifted previous plan
        self.solved = torch.zeros((), dtype=torch.int64, device=dev)
        self.iter_sum = torch.zeros((), dtype=torch.int64, device=dev)
        # the tensors above were filled on the caller's current stream; replan() works on self.stream (a non-blocking
        # torch.cuda.Stream is not ordered behind the default stream): finish that work here, once.  (A host wait, not
        # stream.wait_stream: a set's stream that has once waited on the default stream no longer ran side by side with the
        # other sets' streams on this runtime -- four sets took 28 ms per replan instead of 17.)
        torch.cuda.current_stream(dev).synchronize()

    def _call(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d %s" % (what, rc, self.P.lib.qtos_last_error(self.P.h)))

    def replan(self):
        """One replan of every window: begin() + poll() until the call is queued to its end (results: synchronise the stream)."""
        self.begin()
        self.P.wait()          # (inside the library: the GIL is released, other sets' host threads are not held up)
        self.poll()
        return self.nodes, self.status

    def begin(self):
        """Queue the next replan (hand-over rows, new start / goal vectors, the whole solve) on the set's stream and return
        at once: several sets of windows are kept in flight by ONE host thread that begins / polls them in turn."""
        with self.torch.cuda.stream(self.stream):     # (several window sets may run side by side, each on its own stream)
            self._begin()
        self._pending = True

    def poll(self):
        """True once the replan begun last has been queued to its end (qtos_plan_poll); tallies its results on the stream."""
        if not getattr(self, "_pending", False):
            return True
        if not self.P.poll():
            return False
        with self.torch.cuda.stream(self.stream):
            self.solved.add_((self.status == 0).sum())
            self.iter_sum.add_(self.iters.sum())
        self.have_plan = True
        self._pending = False
        return True

    def _begin(self):
        import ctypes as C
        torch, P, B = self.torch, self.P, self.B
        sp = C.c_void_p(self.stream.cuda_stream)
        warm_ptr = None
        if self.have_plan:
            # hand-over rows: sample the newest plan from `advance` on, take the first row with all feet in contact
            # (force columns 25.. of a foot are non-zero exactly in stance: the reference tests foot heights against
            # the terrain's height set, QTOS/combiner.py:78-92 -- the same rows on these maps)
            n_rows = self.rows.shape[1]
            self._call(P.lib.qtos_sample_csv_device(P.h, B, self.nodes.data_ptr(), self.t0.data_ptr(), C.c_double(self.hz), n_rows,
                                                    self.rows.data_ptr(), sp), "qtos_sample_csv_device")
            k0 = n_rows - self.n_search - 1
            cand = self.rows[:, k0:, :]
            contact = (cand[:, :, 25:37].reshape(B, -1, 4, 3)[..., 2] > 0).all(dim=2)
            first = torch.where(contact.any(dim=1), contact.to(torch.int32).argmax(dim=1), torch.zeros((B,), dtype=torch.int64, device=self.dev))
            idx = k0 + first
            hand = self.rows[torch.arange(B, device=self.dev), idx]
            self.start.copy_(hand[:, 1:25])
            self.offset.copy_(idx.to(torch.float64) / self.hz)
            if self.x_range is not None:
                lo, hi = self.x_range
                x = self.start[:, 0]
                sgn = torch.where(x > hi, -torch.ones_like(x), torch.where(x < lo, torch.ones_like(x), torch.sign(self.goal_step[:, 0])))
                self.goal_step[:, 0] = sgn * self.goal_step[:, 0].abs()
            self.goal[:, 0:2] = self.start[:, 0:2] + self.goal_step[:, 0:2]
            self.nodes, self.prev = self.prev, self.nodes
            if self.warm_mode == "shifted":
                self._call(P.lib.qtos_shift_warm_device(P.h, B, self.prev.data_ptr(), self.offset.data_ptr(), self.start.data_ptr(),
                                                        self.goal.data_ptr(), None if self.map_id is None else self.map_id.data_ptr(),
                                                        self.warm.data_ptr(), sp), "qtos_shift_warm_device")
                mix = getattr(self, "_mix", "all")
                if mix != "all":     # (scratch/shift_exp.py: only part of the shifted plan, the rest from the straight-line guess)
                    guess = torch.empty_like(self.warm)
                    big = torch.full_like(self.offset, 1e9)
                    self._call(P.lib.qtos_shift_warm_device(P.h, B, self.prev.data_ptr(), big.data_ptr(), self.start.data_ptr(),
                                                            self.goal.data_ptr(), None if self.map_id is None else self.map_id.data_ptr(),
                                                            guess.data_ptr(), sp), "qtos_shift_warm_device")
                    nbn = P.dims.n_base_nodes
                    if mix == "cold":
                        self.warm.copy_(guess)
                    elif mix == "base":
                        self.warm[:, 12 * nbn:] = guess[:, 12 * nbn:]
                    elif mix == "base+feet":
                        self.warm[:, self._force_off:] = guess[:, self._force_off:]
                warm_ptr = self.warm.data_ptr()
        self._call(P.lib.qtos_plan_submit(P.h, B, self.start.data_ptr(), self.goal.data_ptr(),
                                          None if self.map_id is None else self.map_id.data_ptr(), warm_ptr,
                                          self.nodes.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(),
                                          self.viol.data_ptr(), sp), "qtos_plan_submit")
