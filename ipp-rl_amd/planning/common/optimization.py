"""
simulate_prediction_step: the env-step API every reference planner calls (planning/common/optimization.py:
14-30), as ONE fused HIP call: mask -> covariance-only predict -> masked trace-reduction reward.
"""
from typing import Dict, Tuple

import numpy as np

from ... import _runtime


def simulate_prediction_step(
    current_state: np.array,
    previous_action: np.array,
    action: np.array,
    mapping,
    uav_specifications: Dict = None,
    adaptive_info: Dict = None,
) -> Tuple[float, np.array, np.array]:
    eng, _ = mapping._engine()
    eng.write_cov(0, current_state)
    if adaptive_info is not None:
        eng.write_mean(0, adaptive_info["mean"])
        eng.set_adaptive(adaptive_info["value_threshold"], adaptive_info["interval_factor"])
    if uav_specifications is not None:
        eng.set_uav(uav_specifications["max_v"], uav_specifications["max_a"])
    a = np.asarray(action, dtype=np.float64).reshape(1, 3)
    p = np.asarray(previous_action, dtype=np.float64).reshape(1, 3)
    reward, status = eng.step(a, p, env_ids=[0], cov_only=True, adaptive=adaptive_info is not None,
                              use_flight_time=uav_specifications is not None)
    if int(status[0]) not in (0, 1):
        raise ValueError(f"HIP step rejected the action (status {int(status[0])})")
    return float(reward[0]), action, _runtime.to_host64(eng.read_cov(0))
