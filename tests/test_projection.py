"""The reduced base's spline space and the projection of given nodes onto it, against numpy / scipy (oracle/projection.py:
scipy.interpolate.BSpline for the map Z, the polar form of a cubic for the coefficients) -- not against the product's own
tables.  `-m gpu`: qtos_project_nodes runs the product's kernel."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401  (path set-up)


def _setup(gait="walk"):
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100(gait=gait)
    P = Planner(cfg, max_batch=8)
    O = Oracle(oracle_dict(cfg))
    _, vf, _ = P.structure()
    return cfg, P, O, vf


@pytest.mark.gpu
@pytest.mark.parametrize("gait", ["walk", "trot"])
def test_projection_is_the_identity_on_the_spline_space_and_z_is_scipys_bspline(gait):
    """Random coefficients -> node values and velocities by scipy's BSpline on the knot vector with the two double knots ->
    the product's projection returns them unchanged (1e-12): its coefficient rule recovers the coefficients and its map
    `nodes = Z c` is scipy's evaluation of the spline and of its derivative."""
    from oracle.projection import base_knots, nodes_of_coefficients
    cfg, P, O, vf = _setup(gait)
    nb = O.L.n_base_nodes - 1
    knots, t = base_knots(nb, O.L.T)
    assert len(knots) - 4 == nb + 5 and P.dims.n_unknowns < P.dims.n_free + P.dims.n_eq_work     # (the reduced base is on)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(8, P.n))
    for row in x:
        for off in (O.L.off_lin, O.L.off_ang):
            for d in range(3):
                pv, vv = nodes_of_coefficients(knots, rng.normal(size=len(knots) - 4), t)
                row[off + 6 * np.arange(nb + 1) + d] = pv
                row[off + 6 * np.arange(nb + 1) + 3 + d] = vv
    # (reduce_swing, the default: given nodes also get their swing mid nodes placed on the swing rule -- the oracle's own
    #  restatement of towr's swing_constraint.cc puts these on it beforehand)
    assert O.swing_start_on_rule
    x = np.array([O.project_swings(row) for row in x])
    y = P.project(x)
    P.close()
    assert np.abs(y - x).max() < 1e-12 * max(1.0, np.abs(x).max())


@pytest.mark.gpu
@pytest.mark.parametrize("gait", ["walk", "trot"])
def test_projection_of_arbitrary_nodes_equals_the_numpy_projection(gait):
    """Random Hermite nodes (not in the space): the product's projection equals the numpy one -- coefficient j = the polar
    form of one cubic piece of the Hermite interpolant at (t[j+1], t[j+2], t[j+3]), free node values = scipy's evaluation of
    the resulting spline; fixed node values and every other variable are kept."""
    from oracle.projection import project_nodes
    cfg, P, O, vf = _setup(gait)
    rng = np.random.default_rng(4)
    x = rng.normal(size=(8, P.n))
    y = P.project(x)
    P.close()
    # (reduce_swing: and the swing mid nodes on the swing rule, by the oracle's restatement of it; the two projections touch
    #  disjoint sets of variables)
    z = np.array([O.project_swings(row) for row in project_nodes(x, O.L, vf)])
    assert np.abs(y - z).max() < 1e-12 * max(1.0, np.abs(x).max())
    nb = O.L.n_base_nodes - 1
    base = np.concatenate([np.arange(O.L.off_lin, O.L.off_lin + 6 * (nb + 1)), np.arange(O.L.off_ang, O.L.off_ang + 6 * (nb + 1))])
    swing = np.nonzero(np.abs(np.array([O.project_swings(row) for row in x]) - x).max(axis=0) > 0)[0]
    assert len(swing) % 4 == 0 and len(swing) >= 16          # (four mid-node variables per swing; the walk has 4 x 4 swings)
    other = np.setdiff1d(np.arange(P.n), np.concatenate([base, swing]))
    assert np.array_equal(y[:, other], x[:, other])
    fixed = base[vf[base] == 0]
    assert len(fixed) > 0 and np.array_equal(y[:, fixed], x[:, fixed])
    assert np.abs(y[:, base] - x[:, base]).max() > 1e-3          # (the nodes were not in the space: something moved)
