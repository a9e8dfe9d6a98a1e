"""
Action helpers (reference planning/common/actions.py:8-106): host-side caller glue in NumPy fp64.  Inside the
env step the same cost formula runs in the HIP prologue kernel (csrc/k_prepare.h, IPP_USE_FLIGHT_TIME).
"""
from typing import Dict, List

import numpy as np


def action_costs(action: np.array, previous_action: np.array, uav_specifications: Dict = None) -> float:
    if uav_specifications is None:
        return compute_distance(action, previous_action)
    return compute_flight_time(action, previous_action, uav_specifications)


def compute_distance(action: np.array, previous_action: np.array) -> float:
    return np.linalg.norm(action - previous_action, ord=2)


def _trapezoid_time(dist, max_v, max_a):
    """accelerate - cruise - decelerate; short hops never reach max_v (reference :19-41)."""
    d_acc = np.minimum(0.5 * dist, np.square(max_v) / (2 * max_a))
    return (dist - 2 * d_acc) / max_v + 2 * np.sqrt(2 * d_acc / max_a)


def compute_flight_times(actions: np.array, previous_action: np.array, uav_specifications: Dict = None):
    dists = np.linalg.norm(actions - previous_action, ord=2, axis=1)
    return _trapezoid_time(dists, uav_specifications["max_v"], uav_specifications["max_a"])


def compute_flight_time(action: np.array, previous_action: np.array, uav_specifications: Dict = None) -> float:
    dist = np.linalg.norm(action - previous_action, ord=2)
    return _trapezoid_time(dist, uav_specifications["max_v"], uav_specifications["max_a"])


def get_actions(previous_action, remaining_budget, grid_map, min_altitude, max_altitude, altitude_spacing,
                uav_specifications: Dict = None) -> List:
    """All cell-centre x altitude-level waypoints reachable within the budget, row-major then level (reference :44-66)."""
    levels = int((max_altitude - min_altitude) / altitude_spacing) + 1
    res = grid_map.resolution
    out = []
    for i in range(grid_map.y_dim):
        for j in range(grid_map.x_dim):
            for k in range(levels):
                a = np.array([res * j + 0.5 * res, res * i + 0.5 * res, min_altitude + altitude_spacing * k])
                if 0 < action_costs(a, previous_action, uav_specifications) <= remaining_budget:
                    out.append(a)
    return out


def flatten_grid_index(grid_map, index_2d: np.array) -> int:
    return int(grid_map.x_dim * index_2d[0] + index_2d[1])


def enumerate_actions(grid_map, min_altitude: float, max_altitude: float, altitude_spacing: float):
    """dict idx -> [x, y, altitude] with idx = level * N + x_dim * col + row (reference :72-92, quirk kept)."""
    levels = np.linspace(min_altitude, max_altitude, int((max_altitude - min_altitude) / altitude_spacing) + 1)
    res = grid_map.resolution
    xs, ys = np.meshgrid(np.arange(grid_map.x_dim) * res, np.arange(grid_map.y_dim) * res)
    offset = np.array([0.5 * res, 0.5 * res], dtype=np.float64)
    positions = np.array([xs.ravel(), ys.ravel()], dtype=np.float64).T + offset
    actions = {}
    for lvl, altitude in enumerate(levels):
        for pos in positions:
            idx = flatten_grid_index(grid_map, (pos - offset) / res)
            actions[lvl * grid_map.num_grid_cells + idx] = np.array([pos[0], pos[1], altitude])
    return actions


def action_dict_to_np_array(actions: Dict) -> np.array:
    arr = np.zeros((len(actions), 3))
    for idx, action in actions.items():
        arr[idx, :] = action
    return arr


def out_of_bounds(waypoint, grid_map, min_altitude: float, max_altitude: float):
    ok_x = 0 <= waypoint[1] <= grid_map.x_dim * grid_map.resolution
    ok_y = 0 <= waypoint[0] <= grid_map.y_dim * grid_map.resolution
    ok_z = min_altitude <= waypoint[2] <= max_altitude
    return not (ok_x and ok_y and ok_z)
