"""Terrain files: the side channel between QTOS's map generator and the solver.

Counterpart of QTOS/generateHeightField.py: tile reader (:100-118, tiles are read transposed),
nearest-neighbour upsampling (:39-56), tile concatenation along x (:470-484), the solver's copy
(transpose, shifted one row toward +x: :568,620-631) and the text format
(``"v, v, ..., v,"`` per line, no final newline: :590-605), and the height-level randomiser
(:692-730).
World frame: map index (row = y, col = x), cell = 2 / rows metres, origin shift 1.0 in x and y
(QTOS/planner.py:61-62), i.e. x in [-1, 2*tiles - 1], y in [-1, 1].
"""
import numpy as np


def read_tile(path, delimiter=','):
    rows = []
    with open(path) as f:
        for line in f.readlines():
            vals = []
            for tok in line.strip().split(delimiter):
                try:
                    vals.append(float(tok))
                except ValueError:
                    pass
            rows.append(vals)
    return np.transpose(np.array(rows))


def scale_map(tile, scale_factor=1):
    return np.repeat(np.repeat(np.asarray(tile), scale_factor, axis=0), scale_factor, axis=1)


def build_map(tiles, mesh_scale=1):
    """List of tile arrays (as read_tile returns them) -> map[y_index][x_index]."""
    return np.concatenate([scale_map(t, mesh_scale) for t in tiles], axis=1)


def towr_map(map_yx):
    """The array the reference writes for the solver: [x_index][y_index], shifted one row in +x."""
    m = np.transpose(np.array(map_yx, dtype=float))
    out = np.zeros_like(m)
    out[1:] = m[:-1]
    return out


def random_height(map_yx, rng, height_delta=0.005):
    """One pass of the reference's terrain randomiser (QTOS/generateHeightField.py:708-730): every
    distinct non-zero height level (ascending) moves as a whole by +d, -d or not at all, d uniform
    in [-height_delta, height_delta].  `rng` is a `random.Random`; seeded like the module-level
    generator the reference uses it reproduces the reference's maps (tests/golden/random_height.json)."""
    out = np.array(map_yx, dtype=float)
    for h in np.unique(out[out != 0]):
        d = rng.uniform(-height_delta, height_delta)
        n = rng.choice((0, 1, 2))
        if n == 0:
            out[out == h] += d
        elif n == 1:
            out[out == h] -= d
    return out


def random_height_shift(map_yx, shift, rng):
    """`shift` cumulative passes of random_height (QTOS/generateHeightField.py:692-706; the
    reference's randomize_env applies 10)."""
    out = np.array(map_yx, dtype=float)
    for _ in range(shift):
        out = random_height(out, rng)
    return out


def cell_size(map_yx):
    return 1.0 / (np.asarray(map_yx).shape[0] / 2.0)


def write_height_file(path, arr):
    rows = len(arr)
    with open(path, 'w') as f:
        for k, line in enumerate(arr):
            f.write(', '.join(str(v) for v in line) + ',')
            if k < rows - 1:
                f.write('\n')


def read_height_file(path):
    """Solver-side reader of the same format -> float array [x_index][y_index]."""
    rows = []
    with open(path) as f:
        for line in f.read().split('\n'):
            vals = [float(t) for t in line.split(',') if t.strip() != '']
            if vals:
                rows.append(vals)
    return np.array(rows)


def height_at(height_xy, cell, x, y, x0=-1.0, y0=-1.0, mode=0):
    """Terrain height as the planner kernels evaluate it: mode 0 bilinear (clamped at the border),
    mode 1 nearest cell."""
    h = np.asarray(height_xy, float)
    fx = np.clip((np.asarray(x, float) - x0) / cell, 0, h.shape[0] - 1)
    fy = np.clip((np.asarray(y, float) - y0) / cell, 0, h.shape[1] - 1)
    if mode == 1:
        return h[np.floor(fx + 0.5).astype(int), np.floor(fy + 0.5).astype(int)]
    ix = np.clip(np.floor(fx).astype(int), 0, max(h.shape[0] - 2, 0))
    iy = np.clip(np.floor(fy).astype(int), 0, max(h.shape[1] - 2, 0))
    ix1 = np.minimum(ix + 1, h.shape[0] - 1)
    iy1 = np.minimum(iy + 1, h.shape[1] - 1)
    u, v = fx - ix, fy - iy
    return (h[ix, iy] * (1 - u) * (1 - v) + h[ix1, iy] * u * (1 - v)
            + h[ix, iy1] * (1 - u) * v + h[ix1, iy1] * u * v)
