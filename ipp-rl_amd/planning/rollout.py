"""
Rollout / expansion policies of the classic MCTS planner (reference planning/mcts_mission.py:167-272) on the batched
engine.  The reference scores every reachable action from a search node with one dense covariance update each
(greedy_action, :232-246; gcb_policy, :204-221) -- here one ipp_tree_score_actions call reads the node's state once for
all candidates (csrc/k_score.h on the chained tree state, csrc/k_tree.h); node states stay on the device as path-local
factor columns (planning/tree.py).  Random draws follow the reference's call sequence (np.random.uniform / choice).
"""
from typing import Dict, Optional, Sequence

import numpy as np

from .common.actions import compute_flight_times


class RolloutPolicy:
    def __init__(self, engine, actions_np: np.ndarray, uav_specifications: Optional[Dict], max_greedy_radius: float,
                 adaptive: bool = True, rng=np.random):
        """engine: factor-state IPPEngine with score_scratch=True and node_capacity > 0."""
        self.engine, self.actions_np, self.uav = engine, np.asarray(actions_np, dtype=np.float64), uav_specifications
        self.max_greedy_radius, self.adaptive, self.rng = float(max_greedy_radius), adaptive, rng

    def next_actions_mask(self, position, budget: float, uav_specification: Optional[Dict] = None) -> np.ndarray:
        """mcts_mission.py:167-173."""
        d = np.linalg.norm(self.actions_np - position, ord=2, axis=1)
        if uav_specification is None:
            return (d > 0) & (d <= budget) & (d < self.max_greedy_radius)
        t = compute_flight_times(self.actions_np, position, uav_specification)
        return (t > 0) & (t <= budget) & (d < self.max_greedy_radius)

    def score(self, root: int, path: Sequence[int], previous_action, actions) -> np.ndarray:
        """compute_reward(node.state, prediction_step(node.state, a), node.action, a) for every candidate a (:235-241)."""
        reward, status = self.engine.tree_score_actions(root, path, actions, previous_action, adaptive=self.adaptive,
                                                        use_flight_time=self.uav is not None)
        if int(status.abs().sum()) != 0:
            raise ValueError("a candidate footprint was rejected by the engine")
        return reward.detach().cpu().numpy().astype(np.float64)

    def greedy_action(self, root: int, path: Sequence[int], previous_action, actions) -> np.ndarray:
        """:232-246: the FIRST maximiser (strict > in the reference's loop)."""
        return np.asarray(actions)[int(np.argmax(self.score(root, path, previous_action, actions)))]

    def eps_greedy_policy(self, root: int, path: Sequence[int], previous_action, remaining_budget: float, epsilon: float):
        """:248-256."""
        msk = self.next_actions_mask(previous_action, remaining_budget, self.uav)
        available = self.actions_np[msk]
        if self.rng.uniform(0, 1) > epsilon and msk.sum() > 0:
            return self.greedy_action(root, path, previous_action, available)
        return available[self.rng.choice(len(available))]

    def gcb_policy(self, root: int, path: Sequence[int], previous_action, remaining_budget: float):
        """:204-221: sample an action with softmax(benefit-to-cost) probabilities."""
        available = self.actions_np[self.next_actions_mask(previous_action, remaining_budget, self.uav)]
        v = self.score(root, path, previous_action, available)
        p = np.exp(v) / np.sum(np.exp(v))
        return available[self.rng.choice(len(available), p=p)]

    @staticmethod
    def widen(num_children: int, visits: int, k: float, alpha: float, num_available: int) -> bool:
        """Progressive widening rule (:263-272): expand a new child instead of selecting an existing one?"""
        return num_children == 0 or (num_children <= k * visits ** alpha and num_children < num_available)
