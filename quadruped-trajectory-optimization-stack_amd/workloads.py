"""Synthetic planning workloads of BASELINE.json: seeded start/goal/terrain batches."""
import os

import numpy as np

from . import heightfield

NOMINAL_FEET = np.array([[0.21, 0.19, 0.0], [0.21, -0.19, 0.0], [-0.21, 0.19, 0.0], [-0.21, -0.19, 0.0]])
TILE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "heightfields")


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
