"""Synthetic planning workloads of BASELINE.json: seeded start/goal/terrain batches."""
import os

import numpy as np

from . import heightfield

NOMINAL_FEET = np.array([[0.21, 0.19, 0.0], [0.21, -0.19, 0.0], [-0.21, 0.19, 0.0], [-0.21, -0.19, 0.0]])
# the 20 x 20 terrain tiles of the reference experiments the workloads are built from (inputs: data/heightfields/*.txt
# of the reference, SURVEY.md section 2 "Heightfields"); tests/golden/heightfields holds the same files as fixtures
TILE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "heightfields")


def rest_start(x0, y0=0.0, z=0.24, feet_z=None):
    """Start vector (24): CoM, Euler, feet FL FR HL HR, velocity, Euler rates -- robot at rest."""
    feet = NOMINAL_FEET + np.array([x0, y0, 0.0])
    if feet_z is not None:
        feet[:, 2] = feet_z
    return np.concatenate([[x0, y0, z], [0, 0, 0], feet.ravel(), [0, 0, 0], [0, 0, 0]])


def flat_goals(batch, seed=0):
    """configs[1]: exp_1 flat ground, start (x0,0,0.24) x0~U(0,2), goal = start + (U(.3,.6), U(-.05,.05))."""
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(0.0, 2.0, batch)
    dx = rng.uniform(0.3, 0.6, batch)
    dy = rng.uniform(-0.05, 0.05, batch)
    start = np.stack([rest_start(x) for x in x0])
    goal = np.stack([x0 + dx, dy, np.full(batch, 0.24)], axis=1)
    return start, goal


def tile(name):
    return heightfield.read_tile(os.path.join(TILE_DIR, name + ".txt"))


def exp5_terrain(mesh_scale=11):
    """exp_5: tiles climb_2, climb_1 (data/config/experiment_5_extreme_climbing.yml:4,6) in the
    solver file's orientation [x][y]; returns (height_xy, cell)."""
    m = heightfield.build_map([tile("climb_2"), tile("climb_1")], mesh_scale)
    return heightfield.towr_map(m), heightfield.cell_size(m)


def exp1_terrain():
    m = heightfield.build_map([tile("plane"), tile("plane")], 1)
    return heightfield.towr_map(m), heightfield.cell_size(m)


def step_goals(batch, seed=1, terrain=None, mode=1):
    """configs[2]: starts on the flat part x in [0, 0.2], goals stepping onto the 0.025/0.05 m
    ledges at x ~ 0.3-0.5 (tile rows 13-19 of climb_2); feet start on the terrain surface."""
    rng = np.random.default_rng(seed)
    height_xy, cell = terrain if terrain is not None else exp5_terrain()
    x0 = rng.uniform(0.0, 0.2, batch)
    dx = rng.uniform(0.30, 0.45, batch)
    dy = rng.uniform(-0.03, 0.03, batch)
    start = []
    for x in x0:
        feet = NOMINAL_FEET + np.array([x, 0.0, 0.0])
        fz = heightfield.height_at(height_xy, cell, feet[:, 0], feet[:, 1], mode=mode)
        start.append(rest_start(x, 0.0, 0.24 + float(heightfield.height_at(height_xy, cell, x, 0.0, mode=mode)), fz))
    goal = np.stack([x0 + dx, dy, np.full(batch, 0.24)], axis=1)
    return np.stack(start), goal


def mixed_terrains(mesh_scale=11):
    """configs[3]: exp_1 / exp_3 / exp_5 heightfields on ONE common grid (the planner takes a stack of
    equally shaped maps + a map index per problem).  All three are built at mesh_scale 11 -- the
    reference upsamples by replication, so the coarser experiments are represented exactly -- and
    padded with flat ground to exp_3's length.  Returns (maps[3][nx][ny], cell)."""
    e1 = heightfield.build_map([tile("plane"), tile("plane")], mesh_scale)
    e3 = heightfield.build_map([tile("feasibility_test"), tile("feasibility_test_1"), tile("plane")], mesh_scale)
    e5 = heightfield.build_map([tile("climb_2"), tile("climb_1")], mesh_scale)
    nx = max(m.shape[1] for m in (e1, e3, e5))
    out = []
    for m in (e1, e3, e5):
        t = heightfield.towr_map(m)
        pad = np.zeros((nx, t.shape[1]))
        pad[:t.shape[0]] = t
        out.append(pad)
    return np.stack(out), heightfield.cell_size(e1)


def mixed_goals(batch, seed=2, terrains=None):
    """A third each of exp_1 flat goals, exp_3 corridor goals (y = 0 lane between the blocks, x in
    [0, 3.5]) and exp_5 step goals; returns (start, goal, map_id)."""
    maps, cell = terrains if terrains is not None else mixed_terrains()
    rng = np.random.default_rng(seed)
    n1 = batch // 3
    n3 = batch // 3
    n5 = batch - n1 - n3
    s1, g1 = flat_goals(n1, seed)
    # exp_3: the robot walks in the free lane between the 0.5 m blocks: start and goal stances are
    # drawn (rejection sampling) where all four feet stand on the floor
    s3, g3 = [], []
    while len(s3) < n3:
        x = rng.uniform(0.0, 3.5)
        dx, dy = rng.uniform(0.3, 0.5), rng.uniform(-0.03, 0.03)
        feet = NOMINAL_FEET + np.array([x, 0.0, 0.0])
        feet_goal = NOMINAL_FEET + np.array([x + dx, dy, 0.0])
        pts = np.concatenate([feet, feet_goal, 0.5 * (feet + feet_goal)])
        if np.any(heightfield.height_at(maps[1], cell, pts[:, 0], pts[:, 1], mode=1) != 0.0):
            continue
        s3.append(rest_start(x, 0.0, 0.24, np.zeros(4)))
        g3.append([x + dx, dy, 0.24])
    g3 = np.array(g3)
    s5, g5 = step_goals(n5, seed + 1, terrain=(maps[2], cell))
    start = np.concatenate([s1, np.stack(s3), s5])
    goal = np.concatenate([g1, g3, g5])
    map_id = np.concatenate([np.zeros(n1, np.int32), np.ones(n3, np.int32), np.full(n5, 2, np.int32)])
    perm = rng.permutation(batch)
    return start[perm], goal[perm], map_id[perm]


def random_terrains(n_maps=8, seed=4, mesh_scale=11, shift=10):
    """configs[4]: randomized heightfields -- the exp_5 map with every height level jittered by the
    reference's own randomiser (`random_height_shift`, 10 cumulative passes of +-0.005 m per level as in
    its `randomize_env`, QTOS/generateHeightField.py:560-567,692-730), one `random.Random(seed + m)`
    stream per map, in the solver file's orientation.  Returns (maps[n_maps][nx][ny], cell)."""
    import random
    base = heightfield.build_map([tile("climb_2"), tile("climb_1")], mesh_scale)
    maps = [heightfield.towr_map(heightfield.random_height_shift(base, shift, random.Random(seed + m)))
            for m in range(n_maps)]
    return np.stack(maps), heightfield.cell_size(base)


def mpc_goals(batch, seed=5, terrains=None, mode=1):
    """configs[4]: long-horizon goals (0.45-0.7 m ahead over the 10 s / two-cycle horizon: up the
    randomized ledges) from start stances at x in [0, 0.2] standing on the surface of their map, each
    problem on one of the randomized heightfields; returns (start, goal, map_id)."""
    maps, cell = terrains if terrains is not None else random_terrains()
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(0.0, 0.2, batch)
    dx = rng.uniform(0.45, 0.7, batch)
    dy = rng.uniform(-0.03, 0.03, batch)
    map_id = rng.integers(0, len(maps), batch).astype(np.int32)
    start = []
    for x, m in zip(x0, map_id):
        feet = NOMINAL_FEET + np.array([x, 0.0, 0.0])
        fz = heightfield.height_at(maps[m], cell, feet[:, 0], feet[:, 1], mode=mode)
        start.append(rest_start(x, 0.0, 0.24 + float(heightfield.height_at(maps[m], cell, x, 0.0, mode=mode)), fz))
    goal = np.stack([x0 + dx, dy, np.full(batch, 0.24)], axis=1)
    return np.stack(start), goal, map_id
