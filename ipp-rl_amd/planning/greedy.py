"""
Greedy one-step planner on the batched engine (config 1 plumbing; counterpart of the reference's
planning/common/optimization.py:33-104 greedy_search and planning/greedy_mission.py:73-110 loop).

The reference scores every reachable action by pickling the Mapping and its 50 MB covariance into a
multiprocessing.Pool(4) and running simulate_prediction_step per candidate; here all candidates are scored by
ONE ipp_score_actions call that reads the state once (csrc/k_score.h), the winner is committed in place.
``shared_pass=False`` scores through a predict-only batched ipp_step instead (env id repeated), which streams the
state once per candidate.
"""
from typing import Dict, List, Optional

import numpy as np

from ..engine import EngineConfig, IPPEngine
from .common.actions import action_costs, get_actions

INIT_ACTION = np.array([2.0, 2.0, 14.0])  # reference planning/missions.py:69


class GreedyPlanner:
    def __init__(self, cfg: EngineConfig, min_altitude: float, max_altitude: float, altitude_spacing: float,
                 uav_specifications: Optional[Dict] = None, adaptive: bool = True, state: str = "dense",
                 device: str = "cuda:0", max_candidates: Optional[int] = None, shared_pass: bool = True):
        levels = int((max_altitude - min_altitude) / altitude_spacing) + 1
        self.cfg = cfg
        self.min_altitude, self.max_altitude, self.altitude_spacing = min_altitude, max_altitude, altitude_spacing
        self.uav = uav_specifications
        self.adaptive = adaptive
        cap = max_candidates or cfg.n_cells * levels
        self.shared_pass = shared_pass
        self.engine = IPPEngine(cfg, capacity=2, state=state, rank_cap=9 * 128, max_batch=max(cap, 2), device=device,
                                score_scratch=shared_pass)
        if uav_specifications is not None:
            self.engine.set_uav(uav_specifications["max_v"], uav_specifications["max_a"])

    class _Grid:
        def __init__(self, cfg):
            self.x_dim, self.y_dim, self.resolution = cfg.x_dim, cfg.y_dim, cfg.resolution
            self.num_grid_cells = cfg.n_cells

    def candidates(self, previous_action, remaining_budget) -> List[np.ndarray]:
        return get_actions(previous_action, remaining_budget, self._Grid(self.cfg), self.min_altitude, self.max_altitude,
                           self.altitude_spacing, self.uav)

    def score(self, previous_action, candidates, env: int = 0) -> np.ndarray:
        """Reward of every candidate from the current state of slot `env` (nothing is written)."""
        acts = np.asarray(candidates, dtype=np.float64).reshape(-1, 3)
        if self.shared_pass:
            reward, _ = self.engine.score_actions(env, acts, previous_action, adaptive=self.adaptive,
                                                  use_flight_time=self.uav is not None)
            return reward.detach().cpu().numpy().astype(np.float64)
        prev = np.tile(np.asarray(previous_action, dtype=np.float64), (len(acts), 1))
        ids = np.full(len(acts), env, dtype=np.int32)
        reward, _ = self.engine.step(acts, prev, env_ids=ids, cov_only=True, predict_only=True, adaptive=self.adaptive,
                                     use_flight_time=self.uav is not None)
        return reward.detach().cpu().numpy().astype(np.float64)

    def search(self, previous_action, remaining_budget: float, episode_horizon: int) -> List[np.ndarray]:
        """greedy_search (planning/common/optimization.py:33-104): `episode_horizon` waypoints, each the first
        maximiser of the predicted reward from the look-ahead state; the look-ahead runs on scratch slot 1
        (covariance-only commits, the map mean and the adaptive mask's mean stay those of slot 0)."""
        prev = np.asarray(previous_action, dtype=np.float64)
        budget = float(remaining_budget)
        self.engine.fork([0], [1])
        waypoints = []
        for _ in range(episode_horizon):
            cands = self.candidates(prev, budget)
            if len(cands) == 0:
                break
            r = self.score(prev, cands, env=1)
            best = np.asarray(cands[int(np.argmax(r))])
            self.engine.step(best[None], prev[None], env_ids=[1], cov_only=True, adaptive=self.adaptive,
                             use_flight_time=self.uav is not None)
            budget -= action_costs(best, prev, self.uav)
            prev = best
            waypoints.append(best)
        return waypoints

    def reset(self, white_noise=None, gt=None, prior_scale=None):
        self.engine.reset(env_ids=[0], white_noise=None if white_noise is None else np.asarray(white_noise)[None],
                          gt=None if gt is None else np.asarray(gt)[None], prior_scale=prior_scale)

    def run(self, budget: float, meas_noise_fn=None, previous_action=None, trace: Optional[list] = None):
        """Greedy mission: score -> argmax -> execute (observe + update) until the budget is spent.
        meas_noise_fn(m) returns the standard normals of one measurement (NumPy legacy stream for parity).
        trace: a list that receives (candidates, rewards) of every step (tests)."""
        prev = INIT_ACTION.copy() if previous_action is None else np.asarray(previous_action, dtype=np.float64)
        waypoints, rewards = [], []
        while budget >= 0:
            cands = self.candidates(prev, budget)
            if len(cands) == 0:
                break
            r = self.score(prev, cands)
            if trace is not None:
                trace.append((np.asarray(cands), r))
            best = np.asarray(cands[int(np.argmax(r))])
            eps = None
            if meas_noise_fn is not None:
                eps = np.zeros((1, self.engine.meas_cap))
                e = np.ravel(meas_noise_fn(best))
                eps[0, : e.size] = e
            self.engine.step(best[None], prev[None], env_ids=[0], meas_noise=eps, adaptive=self.adaptive,
                             use_flight_time=self.uav is not None)
            budget -= action_costs(best, prev, self.uav)
            prev = best
            waypoints.append(best)
            rewards.append(float(np.max(r)))
        return np.array(waypoints), np.array(rewards), budget
