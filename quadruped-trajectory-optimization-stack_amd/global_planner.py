"""Global planner: A* over the boolean map + cubic "spine" + start/goal pair generation.
Counterpart of QTOS/planner.py (``PATH_Solver`` :282-457, ``Global_Planner`` :15-280), pure
numpy/scipy like the reference (not a GPU workload); it produces the batch axis of the local
planner (start / goal per re-plan).  Quirks kept on purpose: the spline's end point is appended
WITHOUT the origin shift (:410,412), pairs are popped LIFO (QTOS/containers.py:171-181).
"""
import heapq
import math

import numpy as np
from scipy.interpolate import CubicSpline


class PathSolver:
    def __init__(self, map_yx, start, goal, step_size=1.0, grid_res=0.1, origin_x_shift=1.0,
                 origin_y_shift=1.0, bool_map=None):
        self.visual_map = np.asarray(map_yx)
        self.bool_map = self.visual_map if bool_map is None else np.asarray(bool_map)
        self.grid_res, self.step_size = grid_res, step_size
        self.origin_x_shift, self.origin_y_shift = origin_x_shift, origin_y_shift
        self.solution_flag = False
        self.start_idx = self.convert_2_idx(start[0], start[1])
        self.goal_idx = self.convert_2_idx(goal[0], goal[1])
        self.path = self.astar(self.start_idx, self.goal_idx)
        self.predicted_t = np.linalg.norm(np.array(start[0:2]) - np.array(goal[0:2])) / step_size * 10
        if self.solution_flag:
            self._solve()

    def convert_2_idx(self, x, y):
        return (math.floor((y + self.origin_y_shift) / self.grid_res),
                math.floor((x + self.origin_x_shift) / self.grid_res))

    @staticmethod
    def heuristic(a, b):
        return np.sqrt((b[0] - a[0]) ** 2 + (b[1] - a[1]) ** 2)

    def astar(self, start, goal, height_bound=0.2):
        """4-neighbour A*, Euclidean cost, cells with bool_map > height_bound are blocked
        (QTOS/planner.py:354-399; same tie-breaking: heap of (f, cell))."""
        neighbors = [(0, 1), (0, -1), (1, 0), (-1, 0)]
        close_set, came_from = set(), {}
        gscore = {start: 0}
        oheap = [(self.heuristic(start, goal), start)]
        while oheap:
            current = heapq.heappop(oheap)[1]
            if current == goal:
                data = []
                while current in came_from:
                    data.append(current)
                    current = came_from[current]
                self.solution_flag = True
                return [start] + data[::-1]
            close_set.add(current)
            for i, j in neighbors:
                nb = (current[0] + i, current[1] + j)
                g = gscore[current] + self.heuristic(current, nb)
                if not (0 <= nb[0] < self.bool_map.shape[0] and 0 <= nb[1] < self.bool_map.shape[1]):
                    continue
                if self.bool_map[nb[0]][nb[1]] > height_bound:
                    continue
                if nb in close_set and g >= gscore.get(nb, 0):
                    continue
                if g < gscore.get(nb, 0) or nb not in [e[1] for e in oheap]:
                    came_from[nb] = current
                    gscore[nb] = g
                    heapq.heappush(oheap, (g + self.heuristic(nb, goal), nb))
        return None

    def _solve(self):
        sub = self.path[::2]
        t = np.linspace(0, self.predicted_t, len(sub) + 1)
        xs = [(c[1] * self.grid_res) - self.origin_x_shift for c in sub]
        xs.append(self.path[-1][1] * self.grid_res)          # sic: no origin shift on the end point
        ys = [(c[0] * self.grid_res) - self.origin_y_shift for c in sub]
        ys.append(self.path[-1][0] * self.grid_res)
        self.spine_x_track = CubicSpline(t, xs)
        self.spine_y_track = CubicSpline(t, ys)


class GlobalPlanner:
    def __init__(self, map_yx, start, robot_goal, step_size=1.0, resolution=0.1, lookahead=7500,
                 hz=1000, bool_map=None, history=500):
        self.map = np.asarray(map_yx)
        self.grid_res, self.step_size = resolution, step_size
        self.lookahead, self.hz = lookahead, hz
        self.approx_z = 0.24
        self.origin_x_shift = self.origin_y_shift = 1.0
        self.robot_goal = list(robot_goal)
        self.path_solver = PathSolver(self.map, start, robot_goal, step_size, resolution, bool_map=bool_map)
        self.max_t = self.path_solver.predicted_t
        self._stack, self._cap = [], history

    def convert_2_idx(self, x, y):
        return (math.floor((y + self.origin_y_shift) / self.grid_res),
                math.floor((x + self.origin_x_shift) / self.grid_res))

    def get_map_height(self, pos):
        try:
            row, col = self.convert_2_idx(pos[0], pos[1])
            return self.map[row, col]
        except Exception:
            r, c = self.map.shape
            return self.map[r - 1, c // 2]

    def spine_step(self, com, timestep, total_traj_time=5.0, tol=0.00001):
        tf = timestep + total_traj_time
        sx, sy = self.path_solver.spine_x_track, self.path_solver.spine_y_track
        xf = sx(tf).item() if np.abs(sx(tf)) > tol else 0.0
        yf = sy(tf).item() if np.abs(sy(tf)) > tol else 0.0
        goal = np.array([xf, yf, self.get_map_height((xf, yf)) + self.approx_z])
        return com + np.clip(goal - com, -self.step_size, self.step_size)

    def lookahead_timestamp(self, time):
        return time + round(self.lookahead / self.hz, 3)

    def update(self, timestep):
        """Push one (start, goal) pair for a re-plan issued at ``timestep`` (QTOS/planner.py:195-230)."""
        lt = self.lookahead_timestamp(timestep)
        start = np.array([self.path_solver.spine_x_track(lt), self.path_solver.spine_y_track(lt), 0.0])
        start[2] = self.get_map_height(start[0:2]) + self.approx_z
        goal = self.spine_step(start, lt)
        if len(self._stack) >= self._cap:
            self._stack.pop(0)
        self._stack.append((start, goal))

    def pop(self):
        return self._stack.pop()   # LIFO, like Limited_Stack

    def empty(self):
        return not self._stack
