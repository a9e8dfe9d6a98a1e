"""
NN input feature planes on the HIP engine (reference planning/common/features.py).

Same functions and arguments as the reference.  The N x N covariance planes -- mask rows / columns outside the
adaptive mask, min-max normalise -- are produced by ipp_state_plane on the device; the constant planes (position,
budget) are fills.  Like the reference (features.py:98-99) the states of the history are masked IN PLACE.
This single-env path uploads each fp64 state (compatibility surface); the batched driver calls
``IPPEngine.state_plane`` on env slots instead and never moves a covariance across PCIe.
"""
from typing import Dict, List, Tuple

import numpy as np

from ... import _runtime
from ..._runtime import to_host64
from .actions import action_costs, enumerate_actions
from .rewards import compute_adaptive_msk


class EpisodeHistory:
    def __init__(self, max_history_length: int):
        self.max_history_length = max_history_length
        self.states = []
        self.positions = []
        self.budgets = []

    def push(self, state: np.array, action: np.array, budget: float):
        self.states.insert(0, state)
        self.positions.insert(0, action)
        self.budgets.insert(0, budget)
        if len(self.states) > self.max_history_length:
            self.states.pop()
            self.positions.pop()
            self.budgets.pop()

    def pop(self) -> Tuple[np.array, np.array, float]:
        return self.states.pop(), self.positions.pop(), self.budgets.pop()

    def __len__(self):
        return len(self.states)


def min_max_normalize(x: np.array) -> np.array:
    min_value, max_value = np.min(x), np.max(x)
    if min_value == max_value:
        return x / max_value
    return (x - min_value) / (max_value - min_value)


def generate_position_feature_planes(mapping, position: np.array, min_altitude: float, max_altitude: float):
    n = mapping.grid_map.num_grid_cells
    extent = mapping.grid_map.x_dim * mapping.grid_map.resolution  # the reference divides x AND y by x_dim * resolution
    ones = np.ones((n, n))
    return (position[0] / extent * ones, position[1] / extent * ones,
            (position[2] - min_altitude) / (max_altitude - min_altitude) * ones)


def generate_costs_feature_plane(mapping, current_action: np.array, min_altitude: float, uav_specifications: Dict = None):
    current_action[-1] = min_altitude  # in place, like the reference (features.py:63)
    n = mapping.grid_map.num_grid_cells
    plane = np.zeros((n, n))
    for i, action in enumerate_actions(mapping.grid_map, min_altitude, min_altitude, 1).items():
        plane[i, :] = action_costs(current_action, action, uav_specifications=uav_specifications)
    return min_max_normalize(plane)


def get_field_of_view_indices(mapping, position: np.array) -> List:
    xl, xr, yu, yd = mapping.sensor.project_field_of_view(position)
    xs = np.linspace(xl, xr, int(np.ceil(xr - xl + 1)))[:-1]
    ys = np.linspace(yu, yd, int(np.ceil(yd - yu + 1)))[:-1]
    xm, ym = np.meshgrid(xs, ys)
    idx = np.array([xm.ravel(), ym.ravel()]).T
    return [int(mapping.grid_map.x_dim * p[0] + p[1]) for p in idx]


def generate_fov_feature_plane(mapping, position: np.array) -> np.array:
    n = mapping.grid_map.num_grid_cells
    sel = np.zeros(n, dtype=bool)
    sel[get_field_of_view_indices(mapping, position)] = True
    return np.outer(sel, sel).astype(np.float64)


def _state_plane(mapping, state: np.ndarray, adaptive_info: Dict = None) -> np.ndarray:
    """One N x N state plane on the device (the state's own slot, or slot 1 of the compat engine for host matrices)."""
    eng, _ = mapping._engine()
    if hasattr(state, "device_slot") and state.device_slot(_runtime.state_store(eng)) is not None:
        slot = state.device_slot(_runtime.state_store(eng))  # a state this layer returned: already on the device
    else:
        slot = 1
        eng.write_cov(1, state)
    if adaptive_info is None:
        return to_host64(eng.state_plane(slot, adaptive=False))
    eng.set_adaptive(adaptive_info["value_threshold"], adaptive_info["interval_factor"])
    plane = to_host64(eng.state_plane(slot, mean_for_mask=np.asarray(adaptive_info["mean"], dtype=np.float32).ravel(), adaptive=True))
    msk = compute_adaptive_msk(adaptive_info["mean"], state, adaptive_info["value_threshold"], adaptive_info["interval_factor"])
    state[~msk, :] = 0  # the reference leaves the history masked (features.py:98-99)
    state[:, ~msk] = 0
    return plane


def generate_input_feature_planes(mapping, episode_history: EpisodeHistory, min_altitude: float = None,
                                  max_altitude: float = None, adaptive_info: Dict = None,
                                  uav_specifications: Dict = None, use_action_costs_input: bool = False) -> np.array:
    state_planes = [_state_plane(mapping, st, adaptive_info) for st in episode_history.states]
    zeros = np.zeros_like(state_planes[0])
    budget_planes = [b * np.ones_like(state_planes[0]) for b in episode_history.budgets]
    total = []
    if min_altitude is None or max_altitude is None:
        for i in range(len(episode_history)):
            total.extend([state_planes[i], generate_fov_feature_plane(mapping, episode_history.positions[i]), budget_planes[i]])
        for _ in range(episode_history.max_history_length - len(episode_history)):
            total.extend([zeros] * 3)
        return np.array(total)
    for i in range(len(episode_history)):
        x, y, z = generate_position_feature_planes(mapping, episode_history.positions[i], min_altitude, max_altitude)
        total.extend([state_planes[i], x, y, z, budget_planes[i]])
    for _ in range(episode_history.max_history_length - len(episode_history)):
        total.extend([zeros] * 5)
    if use_action_costs_input:
        total.append(generate_costs_feature_plane(mapping, episode_history.positions[0], min_altitude, uav_specifications))
    return np.array(total)
