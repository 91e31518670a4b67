"""numpy / scipy restatement of the reduced base's spline space (TEST INFRASTRUCTURE, like everything under oracle/).

The product (csrc/model.hpp, "reduced base") replaces the Hermite node values of the base-motion splines inside the KKT
solve by the coefficients of a clamped cubic B-spline on the same knots, with DOUBLE knots at the first and the last
interior junction, and projects given nodes (a warm start) onto that space before a solve (k_start,
qtos_project_nodes).  This file states the same space and the same projection with scipy.interpolate.BSpline and the
polar form (blossom) of a cubic written out in numpy, so that the tests need not take the product's word for either:

  * base_knots(nb, T)            the knot vector
  * nodes_of_coefficients(...)   node values / velocities of a spline given by its coefficients  (= the map Z)
  * project_nodes(x, L, free)    what given nodes become: coefficient j = the blossom of ONE cubic piece of the Hermite
                                 interpolant at the knots (t[j+1], t[j+2], t[j+3]) -- exact for a spline of the space,
                                 whichever piece inside the coefficient's support is taken; for nodes outside the space
                                 the piece decides, and the rule is the product's (model.hpp: the span j+2, else j+1,
                                 j+3, j: the first that is a real interval) --, then the free node values = Z c.

Variable layout of a base spline (towr order, oracle/qtos_oracle.c:87): off + 6 k + 3 q + d, node k, q = 0 position /
1 velocity, dimension d.
"""
import numpy as np
from scipy.interpolate import BSpline


def base_knots(nb, T):
    t = np.linspace(0.0, T, nb + 1)
    kn = [0.0] * 3
    for k in range(nb + 1):
        kn.append(t[k])
        if k == 1 or k == nb - 1:
            kn.append(t[k])                       # double knots: C1 at the first and the last interior junction
    kn += [T] * 3
    return np.array(kn), t


def nodes_of_coefficients(knots, c, t):
    """(values, first derivatives) at the node times t of the cubic B-spline with coefficients c."""
    s = BSpline(knots, np.asarray(c, float), 3, extrapolate=False)
    d = s.derivative()
    te = np.clip(t, knots[0], knots[-1] - 1e-13 * max(1.0, abs(knots[-1])))   # (the right end belongs to the last interval)
    v, dv = s(te), d(te)
    # at the right end evaluate the last polynomial piece exactly
    v[-1] = c[-1]
    h = knots[-1] - knots[-5]
    dv[-1] = 3.0 * (c[-1] - c[-2]) / h
    return v, dv


def _piece_of(knots, j, t):
    """Index of the base polynomial whose blossom gives coefficient j (the product's rule, model.hpp projection tables)."""
    nb = len(t) - 1
    span_of = []
    i = 3
    for k in range(nb):
        while i + 1 < len(knots) and knots[i + 1] <= t[k] + 1e-12:
            i += 1
        span_of.append(i)
    poly_of_span = {s: k for k, s in enumerate(span_of)}
    for i in (j + 2, j + 1, j + 3, j):
        if 3 <= i < len(knots) and i in poly_of_span:
            return poly_of_span[i]
    raise ValueError("no polynomial piece for coefficient %d" % j)


def coefficients_of_nodes(knots, t, p, v):
    """B-spline coefficients of the Hermite nodes (p, v) at times t: blossom of one cubic piece per coefficient."""
    ncj = len(knots) - 4
    c = np.empty(ncj)
    for j in range(ncj):
        k = _piece_of(knots, j, t)
        h, t0 = t[k + 1] - t[k], t[k]
        a0, a1 = p[k], v[k]
        a2 = (3.0 * (p[k + 1] - p[k]) - (2.0 * v[k] + v[k + 1]) * h) / h ** 2
        a3 = (-2.0 * (p[k + 1] - p[k]) + (v[k] + v[k + 1]) * h) / h ** 3
        s1, s2, s3 = knots[j + 1] - t0, knots[j + 2] - t0, knots[j + 3] - t0
        c[j] = a0 + a1 * (s1 + s2 + s3) / 3.0 + a2 * (s1 * s2 + s1 * s3 + s2 * s3) / 3.0 + a3 * s1 * s2 * s3
    return c


def project_nodes(x, L, var_free, T=None):
    """x: node vector(s) [..., n_vars]; L: layout with off_lin, off_ang, n_base_nodes, T; var_free: 1 = free variable.
    Returns the projected copy: free base node values replaced by those of the projected spline, everything else kept."""
    x = np.array(x, dtype=np.float64, copy=True)
    flat = x.reshape(-1, x.shape[-1])
    nb = L.n_base_nodes - 1
    knots, t = base_knots(nb, L.T if T is None else T)
    for row in flat:
        for off in (L.off_lin, L.off_ang):
            for d in range(3):
                ip = off + 6 * np.arange(nb + 1) + d
                iv = ip + 3
                c = coefficients_of_nodes(knots, t, row[ip], row[iv])
                pv, vv = nodes_of_coefficients(knots, c, t)
                fp, fv = var_free[ip] != 0, var_free[iv] != 0
                row[ip[fp]] = pv[fp]
                row[iv[fv]] = vv[fv]
    return x
