"""
Waypoint sets and travel costs with the reference's names (planning/common/actions.py), host-side NumPy fp64.
Inside the env step the same cost formula runs in the HIP prologue (csrc/k_prepare.h, IPP_USE_FLIGHT_TIME); here the
candidate sets are built as whole arrays instead of per-waypoint Python loops.
"""
from typing import Dict, List, Optional

import numpy as np


# ----------------------------------------------------------------------------------------------- travel cost
def _trapezoid_time(dist, max_v: float, max_a: float):
    """Accelerate - cruise - decelerate; hops shorter than v^2/a never reach max_v."""
    ramp = np.minimum(0.5 * dist, np.square(max_v) / (2 * max_a))
    return (dist - 2 * ramp) / max_v + 2 * np.sqrt(2 * ramp / max_a)


def compute_distance(action: np.ndarray, previous_action: np.ndarray) -> float:
    return np.linalg.norm(action - previous_action, ord=2)


def compute_flight_time(action: np.ndarray, previous_action: np.ndarray, uav_specifications: Dict = None) -> float:
    return _trapezoid_time(compute_distance(action, previous_action), uav_specifications["max_v"], uav_specifications["max_a"])


def compute_flight_times(actions: np.ndarray, previous_action: np.ndarray, uav_specifications: Dict = None) -> np.ndarray:
    hops = np.linalg.norm(actions - previous_action, ord=2, axis=1)
    return _trapezoid_time(hops, uav_specifications["max_v"], uav_specifications["max_a"])


def action_costs(action: np.ndarray, previous_action: np.ndarray, uav_specifications: Dict = None) -> float:
    """Flight time when UAV limits are given, else the Euclidean distance."""
    if uav_specifications is not None:
        return compute_flight_time(action, previous_action, uav_specifications)
    return compute_distance(action, previous_action)


# ----------------------------------------------------------------------------------------------- waypoint sets
def _altitude_levels(min_altitude: float, max_altitude: float, altitude_spacing: float) -> int:
    return int((max_altitude - min_altitude) / altitude_spacing) + 1


def _cell_centres(grid_map) -> np.ndarray:
    """[y_dim, x_dim, 2] array of (x, y) cell centres in metres."""
    half = 0.5 * grid_map.resolution
    xs = grid_map.resolution * np.arange(grid_map.x_dim) + half
    ys = grid_map.resolution * np.arange(grid_map.y_dim) + half
    return np.stack(np.meshgrid(xs, ys), axis=-1)


def get_actions(previous_action, remaining_budget, grid_map, min_altitude, max_altitude, altitude_spacing,
                uav_specifications: Optional[Dict] = None) -> List[np.ndarray]:
    """Every cell-centre x altitude-level waypoint with 0 < cost <= budget, ordered row, column, level."""
    levels = _altitude_levels(min_altitude, max_altitude, altitude_spacing)
    heights = min_altitude + altitude_spacing * np.arange(levels)
    centres = _cell_centres(grid_map)
    grid = np.empty(centres.shape[:2] + (levels, 3))
    grid[..., :2] = centres[:, :, None, :]
    grid[..., 2] = heights
    candidates = grid.reshape(-1, 3)
    # one scalar cost per waypoint through the same scalar routine the step uses, so the budget test is bit-identical
    costs = np.fromiter((action_costs(a, previous_action, uav_specifications) for a in candidates), dtype=np.float64,
                        count=len(candidates))
    keep = (costs > 0) & (costs <= remaining_budget)
    return [a.copy() for a in candidates[keep]]


def flatten_grid_index(grid_map, index_2d: np.ndarray) -> int:
    """Index convention of the action dictionaries: x_dim * first + second."""
    return int(grid_map.x_dim * index_2d[0] + index_2d[1])


def enumerate_actions(grid_map, min_altitude: float, max_altitude: float, altitude_spacing: float) -> Dict[int, np.ndarray]:
    """{level * N + x_dim * col + row: [x, y, altitude]} (the reference flattens (col, row), a quirk kept here)."""
    heights = np.linspace(min_altitude, max_altitude, _altitude_levels(min_altitude, max_altitude, altitude_spacing))
    centres = _cell_centres(grid_map).reshape(-1, 2)
    cells = (centres - 0.5 * grid_map.resolution) / grid_map.resolution
    table = {}
    for level, altitude in enumerate(heights):
        for (x, y), cell in zip(centres, cells):
            table[level * grid_map.num_grid_cells + flatten_grid_index(grid_map, cell)] = np.array([x, y, altitude])
    return table


def action_dict_to_np_array(actions: Dict) -> np.ndarray:
    out = np.zeros((len(actions), 3))
    out[list(actions.keys())] = np.array(list(actions.values())) if actions else out[:0]
    return out


def out_of_bounds(waypoint, grid_map, min_altitude: float, max_altitude: float) -> bool:
    """True when the waypoint leaves the map rectangle or the altitude band (first coordinate against y, like the reference)."""
    inside = (0 <= waypoint[1] <= grid_map.x_dim * grid_map.resolution
              and 0 <= waypoint[0] <= grid_map.y_dim * grid_map.resolution
              and min_altitude <= waypoint[2] <= max_altitude)
    return not inside
