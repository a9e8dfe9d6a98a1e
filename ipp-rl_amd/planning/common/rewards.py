"""
Host-side reward helpers with the reference's names (planning/common/rewards.py): they only touch the diagonals of
dense matrices the caller already holds; the batched engine evaluates the same masked trace reduction inside its
step kernels without forming the posterior covariance.
"""
from typing import Dict, Optional, Union

import numpy as np

from .actions import action_costs

ArrayOrFloat = Union[float, np.ndarray]


def _diagonal(matrix) -> np.ndarray:
    """diag(P): from the engine's cached diagonal when the state lives on the device (_device_array.DeviceCov)."""
    if hasattr(matrix, "device_slot"):
        return matrix.diagonal()
    return np.einsum("ii->i", matrix)


def compute_adaptive_msk(grid_mean: np.ndarray, grid_covariance: np.ndarray, value_threshold: float,
                         interval_factor: float) -> np.ndarray:
    """Cells whose upper confidence bound mean + k * var reaches the threshold (flat, row-major)."""
    upper_bound = np.ravel(grid_mean) + interval_factor * _diagonal(grid_covariance)
    return upper_bound >= value_threshold


def _trace(matrix: np.ndarray, mask: Optional[np.ndarray]) -> float:
    diagonal = _diagonal(matrix)
    return float(diagonal.sum() if mask is None else diagonal[mask].sum())


def compute_reward(current_state: np.ndarray, next_state: np.ndarray, previous_action: np.ndarray, action: np.ndarray,
                   uav_specifications: Dict = None, adaptive_msk: np.ndarray = None) -> float:
    """(Masked) uncertainty removed by the step per unit of travel cost + 1."""
    gain = _trace(current_state, adaptive_msk) - _trace(next_state, adaptive_msk)
    return gain / (1 + action_costs(action, previous_action, uav_specifications))


def scale_value_target(value: float) -> float:
    """Value-target squashing used by the self-play generators: sqrt(v + 1) - 1."""
    return np.sqrt(1 + value) - 1


def invert_scaled_value_target(value: ArrayOrFloat) -> ArrayOrFloat:
    """Inverse of ``scale_value_target``: (s + 1)^2 - 1."""
    return value * (value + 2)
