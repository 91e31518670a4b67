"""Batched feasibility map: counterpart of QTOS/generateHeightField.py ``PATH_MAP`` (:172-404).

The reference enumerates (start, goal) patches two cells apart next to obstacles, runs ONE solver
process per patch in 32 OS processes and keeps only the exit codes.  Here the patches are
enumerated the same way (``probe_patches`` reproduces ``probe_map`` incl. its rounding and index
walk), solved as ONE GPU batch, and the statuses are stamped into the boolean map with the same
rules: success clears the start / middle / goal cells, failure stamps the diamond-shaped
neighbourhood of the start and goal cells (the reference's "mid" stamp is a no-op, :397-399).
"""
import numpy as np


def neighbors_danger_test(m, ix, iy, sz=1):
    nb = ((sz, 0), (-sz, 0), (0, sz), (0, -sz), (sz, sz), (sz, -sz), (-sz, -sz), (-sz, sz))
    for dx, dy in nb:
        if dx + ix >= m.shape[0] or dx + ix < 0:
            return False
        elif dy + iy >= m.shape[1] or dy + iy < 0:
            return False
        elif m[dx + ix][dy + iy] > 0:
            return True
    return False


def probe_patches(m, multi_map_shift=1, res=0.1, origin_shift=1.0):
    """List of (start_xyz, goal_xyz, start_idx, goal_idx) exactly as ``PATH_MAP.probe_map`` queues them."""
    m = np.asarray(m)
    step = res
    x_start = -res * (m.shape[1] / 2) - res / 2 + ((multi_map_shift - 1) * origin_shift)
    y_start = -res * (m.shape[1] / 2) - res / 2 + ((multi_map_shift - 1) * origin_shift)
    x_goal = -res * (m.shape[1] / 2) + res / 2 + ((multi_map_shift - 1) * origin_shift)
    y_goal = -res * (m.shape[1] / 2) - res / 2 + ((multi_map_shift - 1) * origin_shift)
    _x_start, _y_start, _x_goal, _y_goal = x_start, y_start, x_goal, y_goal
    ix, iy, iy2 = 0, 0, 2
    out = []
    for _ in range(m.shape[0]):
        _y_start += step
        _y_goal += step
        _x_start, _x_goal = x_start, x_goal
        for y in range(m.shape[1] // 2 - 1):
            if y == 0:
                _x_start += step
                iy, iy2 = 0, 2
            else:
                _x_start = _x_goal
            _x_goal += 2 * step
            _x_start, _y_start = round(_x_start, 2), round(_y_start, 2)
            _x_goal, _y_goal = round(_x_goal, 2), round(_y_goal, 2)
            if neighbors_danger_test(m, ix, iy) or neighbors_danger_test(m, ix, iy2):
                out.append(((_x_start, _y_start, float(m[ix][iy])), (_x_goal, _y_goal, float(m[ix][iy2])),
                            (ix, iy), (ix, iy2)))
            iy += 2
            iy2 += 2
        ix += 1
    return out


def diamond(scale=1):
    """Cells of the hull of ((-3s,0),(3s,0),(0,-3s),(0,3s)) relative to its centre
    (``find_convex_hull``): |dx| + |dy| <= 3 s, row-major order."""
    r = 3 * scale
    return [(a, b) for a in range(-r, r + 1) for b in range(-r, r + 1) if abs(a) + abs(b) <= r]


def patch_args(start_pt, goal_pt):
    """Solver flags of one patch (``worker_f.state_config``, :365-373)."""
    shift = np.array([start_pt[0], start_pt[1], start_pt[2]])
    return {'-s': [start_pt[0], start_pt[1], start_pt[2] + 0.24],
            '-e1': (np.array([0.21, 0.19, 0.0]) + shift).tolist(),
            '-e2': (np.array([0.21, -0.19, 0.0]) + shift).tolist(),
            '-e3': (np.array([-0.21, 0.19, 0.0]) + shift).tolist(),
            '-e4': (np.array([-0.21, -0.19, 0.0]) + shift).tolist(),
            '-s_ang': [0, 0, 0], '-g': [goal_pt[0], goal_pt[1], goal_pt[2] + 0.24], '-r': 5.0}


def stamp(shape, patches, statuses, scale=1):
    """Exit codes -> boolean map (``worker_f``, :387-404), patches processed in queue order."""
    bm = np.zeros(shape, dtype=int)
    hull = diamond(scale)
    for (_, _, s_idx, g_idx), rc in zip(patches, statuses):
        if rc == 0:
            bm[s_idx] = 0
            bm[s_idx[0], s_idx[1] + 1] = 0
            bm[g_idx] = 0
        else:
            for c in (s_idx, g_idx):
                for a, b in hull:
                    if 0 <= c[0] + a < shape[0] and 0 <= c[1] + b < shape[1]:
                        bm[c[0] + a, c[1] + b] = 1
    return bm


def feasibility_map(local_planner, map_yx, multi_map_shift=1, scale=1):
    """One batched solve over every probe patch -> (bool_map, patches, statuses)."""
    m = np.asarray(map_yx)
    if np.all(m == 0):
        return np.zeros(m.shape, dtype=int), [], []     # check_flat_ground short-cut (:222-225)
    patches = probe_patches(m, multi_map_shift, 0.1 * (1 / scale))
    statuses = local_planner.solve_batch([patch_args(p[0], p[1]) for p in patches], sample=False)
    return stamp(m.shape, patches, statuses, scale), patches, statuses
