"""Several planner handles driven by ONE host thread: the batches of a queue stay in flight on the GPU side by side.

The reference keeps 32 ``docker exec ./main`` processes busy from a queue of probes
(QTOS/generateHeightField.py:344-352: ``while not queue.empty(): data = queue.get()``, 375-377): a slow solve never
holds the other workers.  Here a *lane* is a planner handle with its own workspace and HIP stream; a batch is submitted
to a free lane with ``qtos_plan_submit`` (which queues the start of the solve and returns; include/qtos_planner.h) and
the lanes are polled round-robin with ``qtos_plan_poll``, which reads the counts of unfinished problems an iteration
sends back and queues the next one -- the host drives every Newton iteration, it just never blocks on one: a batch
that waits for its slowest problems keeps a few CUs busy while the next batches run on the others.  No host threads,
nothing blocks but ``drain``.
"""
import ctypes as C


class Lane:
    def __init__(self, planner, stream, B, device):
        import torch
        self.P, self.stream, self.B = planner, stream, B
        f64 = dict(dtype=torch.float64, device=device)
        self.nodes = torch.empty((B, planner.n), **f64)
        self.status = torch.empty((B,), dtype=torch.int32, device=device)
        self.iters = torch.empty((B,), dtype=torch.int32, device=device)
        self.viol = torch.empty((B,), **f64)
        self.tag = None          # what the caller attached to the batch in flight
        self.busy = False


class PlannerPool:
    """``n_lanes`` planner handles of one configuration on one device.

    ``submit(start, goal, map_id=None, warm=None, tag=None)`` queues a batch on a free lane (if every lane is busy it
    first waits for the oldest batch); ``drain()`` waits for everything in flight.  Every finished batch is handed to
    ``on_done(lane)`` exactly once, from inside submit / drain, before its lane is reused: ``lane.nodes / status / iters /
    viol`` (device tensors, first ``lane.n`` rows) and ``lane.tag`` are valid during that call, which runs with the lane's
    stream as the current stream (what it queues is ordered in front of the lane's next batch; accumulate per lane, the
    lanes' streams are not ordered among each other).  Inputs are device tensors (float64 / int32, contiguous), complete
    at submit time, that must stay alive and unchanged until their batch has been handed over."""

    def __init__(self, cfg, n_lanes=3, max_batch=256, device=0, heightfields=None, on_done=None, planner_factory=None):
        import torch
        from .capi import Planner
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.on_done = on_done or (lambda lane: None)
        make = planner_factory or (lambda: Planner(cfg, max_batch=max_batch, device=device))
        self.lanes = []
        for _ in range(n_lanes):
            P = make()
            P.set_kernel_events(False)   # (the pool reads no per-kernel times: capi.Planner.set_kernel_events(True) on a lane's planner turns them back on)
            if heightfields is not None:
                P.set_heightfields(*heightfields)
            self.lanes.append(Lane(P, torch.cuda.Stream(self.dev), max_batch, self.dev))
        self._order = []          # busy lanes, oldest first

    @staticmethod
    def _ptr(t):
        return None if t is None else t.data_ptr()

    def _finish(self, lane):
        self._order.remove(lane)
        with self.torch.cuda.stream(lane.stream):   # whatever on_done queues is ordered in front of the lane's next batch
            self.on_done(lane)
        lane.busy = False

    def _reap(self, block=False):
        """Hand over the batches that are complete; with block, wait for the oldest one (the others are kept fed)."""
        n = 0
        for lane in list(self._order):
            if lane.P.poll() and lane.stream.query():
                self._finish(lane)
                n += 1
        if block and n == 0 and self._order:
            lane = self._order[0]
            while not lane.P.poll():
                for other in self._order[1:]:
                    other.P.poll()
            lane.stream.synchronize()
            self._finish(lane)

    def submit(self, start, goal, map_id=None, warm=None, tag=None, after=None):
        self._reap()
        while all(ln.busy for ln in self.lanes):
            self._reap(block=True)
        lane = next(ln for ln in self.lanes if not ln.busy)
        B = start.shape[0]
        if B < 1 or B > lane.B:
            raise ValueError("batch of %d problems on lanes of %d" % (B, lane.B))
        # The inputs must be complete when they are submitted (synchronise the stream that produced them, or produce them
        # on a stream of your own and pass after=event).  The lanes' streams are never made to wait on the default stream:
        # on this runtime a stream that has once waited on it no longer runs side by side with the others.
        if after is not None:
            lane.stream.wait_event(after)
        lane.P.submit(B, self._ptr(start), self._ptr(goal), self._ptr(map_id), self._ptr(warm), lane.nodes.data_ptr(),
                      lane.status.data_ptr(), lane.iters.data_ptr(), lane.viol.data_ptr(), lane.stream.cuda_stream)
        # (only a submit that went through occupies the lane: one that raised leaves it free)
        lane.tag, lane.busy, lane.n = tag, True, B
        self._order.append(lane)
        return lane

    def drain(self):
        while self._order:
            self._reap(block=True)

    def close(self):
        self.drain()
        for lane in self.lanes:
            lane.P.close()
        self.lanes = []
