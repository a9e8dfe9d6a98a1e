"""
simulate_prediction_step: the env-step API every reference planner calls (planning/common/optimization.py:
14-30), as ONE fused HIP call: mask -> covariance-only predict -> masked trace-reduction reward.

The returned next_state is a DeviceCov (_device_array.py): an array-like that stays on the GPU.  Fed back into the next
call (tree searches chain predictions, planning/mcts_zero/mcts.py:239) it costs neither an upload nor a download; NumPy
functions on it work as on the float64 matrix the reference returns.
"""
from typing import Dict, Tuple

import numpy as np

from ... import _runtime
from ..._device_array import DeviceCov


def simulate_prediction_step(
    current_state: np.array,
    previous_action: np.array,
    action: np.array,
    mapping,
    uav_specifications: Dict = None,
    adaptive_info: Dict = None,
) -> Tuple[float, np.array, np.array]:
    eng, _ = mapping._engine()
    src = _runtime.on_device(eng, current_state)  # a state this layer returned earlier is already there: no upload
    src._pinned = True
    try:
        dst = DeviceCov.new_on_device(_runtime.state_store(eng), src.shape[0])
        s_slot, d_slot = src.device_slot(_runtime.state_store(eng)), dst.device_slot(_runtime.state_store(eng))
        if adaptive_info is not None:
            eng.write_mean(s_slot, adaptive_info["mean"])
            eng.set_adaptive(adaptive_info["value_threshold"], adaptive_info["interval_factor"])
        if uav_specifications is not None:
            eng.set_uav(uav_specifications["max_v"], uav_specifications["max_a"])
        a = np.asarray(action, dtype=np.float64).reshape(1, 3)
        p = np.asarray(previous_action, dtype=np.float64).reshape(1, 3)
        # out of place: P' goes to a new slot, the caller's state is left untouched (mapping/mappings.py:190 allocates)
        reward, status = eng.step(a, p, env_ids=[s_slot], dst_ids=[d_slot], cov_only=True, adaptive=adaptive_info is not None,
                                  use_flight_time=uav_specifications is not None)
        if int(status[0]) not in (0, 1):
            raise ValueError(f"HIP step rejected the action (status {int(status[0])})")
    finally:
        src._pinned = False
    return float(reward[0]), action, dst
