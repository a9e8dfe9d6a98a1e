"""
Reward helpers on host arrays (reference planning/common/rewards.py:8-39).  These read the diagonals of dense
matrices the caller already holds in host memory; the batched engine computes the same masked trace reduction
in-kernel (csrc/k_gain.h) without ever forming P'.
"""
from typing import Dict, Union

import numpy as np

from .actions import action_costs


def compute_adaptive_msk(grid_mean: np.array, grid_covariance: np.array, value_threshold: float, interval_factor: float):
    return grid_mean.flatten(order="C") + interval_factor * np.diag(grid_covariance) >= value_threshold


def compute_reward(current_state, next_state, previous_action, action, uav_specifications: Dict = None,
                   adaptive_msk: np.array = None) -> float:
    before, after = np.diag(current_state), np.diag(next_state)
    if adaptive_msk is not None:
        before, after = before[adaptive_msk], after[adaptive_msk]
    utility = np.sum(before) - np.sum(after)
    return utility / (action_costs(action, previous_action, uav_specifications) + 1)


def scale_value_target(value: float) -> float:
    return np.sqrt(value + 1) - 1


def invert_scaled_value_target(value: Union[float, np.array]) -> Union[float, np.array]:
    return np.square(value) + 2 * value
