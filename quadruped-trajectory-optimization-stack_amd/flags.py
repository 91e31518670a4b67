"""Solver flag marshalling: the process ABI between QTOS and its local planner.

Mirror of QTOS/utils.py:26 (``_flags``) and QTOS/utils.py:644-670 (``cmd_args``): the reference
flattens its ``args`` dict, in insertion order, into ``key value value value `` groups for every
key in ``_flags`` with a truthy value, lists printed with ``str()`` and ``[ ] ,`` stripped.
``parse_flags`` is the inverse the replacement needs on its side of the boundary.

``-r``.  The reference passes ``-r 5.0`` with every feasibility probe (QTOS/generateHeightField.py:373) and
``-r 30 * tiles`` / ``120 * tiles`` next to ``-duration 1.0 * tiles`` / ``4.0 * tiles`` in its default-TOWR mode
(scripts/main.py:119-120, 203-206).  The solver source that reads it is absent from the reference tree and
no committed artefact shows its effect (the logged solves do not carry it), so its meaning is unpinned.
Working hypothesis (the only one consistent with all three call sites): a run-time budget for the solver in
seconds -- short for a probe that only needs an exit status, 30 s of budget per planned second for the long
single solves.  This planner answers in milliseconds, so no such budget can bind: ``-r`` is parsed, range
checked (must be a positive finite number, else ValueError: a malformed flag is rejected loudly) and
reported back in ``LocalPlanner.last["r"]``; it does not enter the NLP.  tests/test_boundary.py holds the
strings; tests/test_gpu_parity.py::test_r_flag_does_not_change_the_plans checks statuses and plans with
and without it.
"""
import numpy as np

FLAGS = ['-g', '-s', '-s_ang', '-s_vel', '-e1', '-e2', '-e3', '-e4', '-t', '-r', '-resolution',
         's_vel', 's_ang_vel', '-duration']
_ARITY = {'-g': 3, '-s': 3, '-s_ang': 3, '-s_vel': 3, '-e1': 3, '-e2': 3, '-e3': 3, '-e4': 3,
          '-t': 1, '-r': 1, '-resolution': 1, 's_vel': 3, 's_ang_vel': 3, '-duration': 1}


def cmd_args(args):
    """Same string the reference appends to ``docker exec <id> ./main``."""
    out = ""
    for key, value in args.items():
        if key in FLAGS and value:
            text = str(value).replace(",", "").replace("[", "").replace("]", "")
            out += key + " " + text + " "
    return out


def parse_flags(argv):
    """Flag string or argv list -> dict with float / list-of-float values."""
    toks = argv.split() if isinstance(argv, str) else list(argv)
    out, i = {}, 0
    while i < len(toks):
        key = toks[i]
        if key not in _ARITY:
            raise ValueError("unknown solver flag %r" % key)
        n = _ARITY[key]
        vals = [float(v) for v in toks[i + 1:i + 1 + n]]
        if len(vals) != n:
            raise ValueError("flag %s expects %d values" % (key, n))
        out[key] = vals if n > 1 else vals[0]
        i += 1 + n
    return out


def problem_arrays(args):
    """args dict -> (start[24], goal[3], t0).  Key meaning: QTOS/combiner.py:166-179."""
    def vec(key, default):
        v = args.get(key)
        if v is None or (hasattr(v, "__len__") and len(v) == 0):
            return list(default)
        return [float(x) for x in v]
    start = (vec('-s', (0, 0, 0.24)) + vec('-s_ang', (0, 0, 0)) + vec('-e1', (0.21, 0.19, 0.0))
             + vec('-e2', (0.21, -0.19, 0.0)) + vec('-e3', (-0.21, 0.19, 0.0))
             + vec('-e4', (-0.21, -0.19, 0.0)) + vec('s_vel', (0, 0, 0)) + vec('s_ang_vel', (0, 0, 0)))
    goal = vec('-g', (0.5, 0.0, 0.24))
    t0 = args.get('-t') or 0.0
    if hasattr(t0, "__len__"):
        t0 = t0[0]
    r = args.get('-r')
    if r is not None and not (isinstance(r, str) and r == ""):
        r = np.ravel(np.asarray(r, dtype=float))     # scalar, list or numpy value alike
    if r is not None and len(r) > 0:
        rv = float(r[0])
        if not (rv > 0.0 and rv < float("inf")):
            raise ValueError("-r must be a positive finite number (see flags.py), got %r" % (r,))
    return start, goal, float(t0)
