"""
Classic Monte Carlo tree search planner on the batched engine: the search loop of the reference's `MCTSMission`
(planning/mcts_mission.py:24-66 Node / UCT, :167-304 policies, progressive widening and `simulate`, :312-389
`run_simulations_proxy`, `merge_roots`, `select_best_child`, `replan`) for MANY roots in lock step.

What the reference does per simulation: descend from the root; at a node that was never visited run a rollout
(eps-greedy or generalised cost-benefit policy, `episode_horizon` deep, discounted by gamma) and return its value; otherwise
either EXPAND a new child (progressive widening: while children <= k * visits^alpha and unexplored actions remain) with the
eps-greedy expansion policy or SELECT a child by UCT, add the edge's reward and recurse.  Every `prediction_step` is a dense
N x N covariance update (`Mapping.update_grid_map(..., cov_only=True, predict_only=True)`) and the greedy policy scores each
reachable action with one more.

Here a search node is the root env slot plus path-local factor columns on the device (ipp_tree_step, csrc/k_tree.h): one
call both creates the child's state and returns the edge's reward `compute_reward(node.state, child.state, node.action,
child.action)` (the reference computes the two separately, from the same arguments); all candidates of a greedy / GCB decision are
scored by ONE ipp_tree_score_actions call that reads the node's state once (planning/rollout.py).  The search of a root is a
generator that yields its device requests; `ClassicMCTS.run` drives the generators of all roots together and puts the
predict steps that are pending at the same time into one ipp_tree_step launch.

Faithful quirks (kept, they define the reference's numbers): UCT's `node.value - min / (max - min)` precedence (:47-52);
children are NOT de-duplicated (the same action can be expanded twice, :258-261); a new child's visit counter is incremented
twice in the simulation that created it (:286-301); `simulate` does not discount (gamma only inside rollouts, :226,:198);
a rollout charges the flight from the GRANDPARENT's waypoint (`previous_action` is `node.action` of the level above, :226);
`max_greedy_radius` is compared with the metric distance (:169-173); every reward of the search is masked with
`compute_adaptive_msk(map mean, node state, value_threshold, interval_factor)` whether or not the mission is adaptive
(:182-187, :207-209, :235-237, :292-297 pass the mask unconditionally; the engine's ipp_set_adaptive / config carries the
two parameters).
Randomness: every root owns a NumPy legacy generator (the reference's `np.random`, seeded `worker_id * 42 + 1` per worker,
:314) and a `random.Random` (the reference's `random.choice` among UCT ties, :66); with the same seeds a root reproduces
the reference's call sequence (tests/test_hip_mcts_mission.py against a recorded search of the imported reference).
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence

import numpy as np

from .common.actions import action_costs, action_dict_to_np_array, enumerate_actions
from .rollout import RolloutPolicy
from .tree import TreeNodePool


class _Grid:
    def __init__(self, cfg):
        self.x_dim, self.y_dim, self.resolution = cfg.x_dim, cfg.y_dim, cfg.resolution
        self.num_grid_cells = cfg.n_cells


class Node:
    """planning/mcts_mission.py:24-66; `dev` is the device node that holds the state (-1: the root env slot itself)."""

    __slots__ = ("parent", "action", "value_sum", "visits", "children", "dev", "path", "root", "edge_reward")

    def __init__(self, root: int, dev: int, path: List[int], parent=None, action=None, edge_reward: float = 0.0):
        self.root, self.dev, self.path = root, dev, path
        self.parent, self.action = parent, np.asarray(action, dtype=np.float64)
        self.value_sum, self.visits, self.children = 0, 0, []
        self.edge_reward = edge_reward  # compute_reward(parent.state, state, parent.action, action): fixed per edge

    @property
    def value(self):
        return self.value_sum / self.visits

    @staticmethod
    def uct(node, min_val: float, max_val: float, c: float = 2.0):
        if node.visits == 0:
            return np.inf
        exploration = c * np.sqrt(np.log(node.parent.visits) / node.visits)
        if max_val == 0:
            return node.value + exploration
        if max_val == min_val:
            normalized_value = node.value / max_val
        else:
            normalized_value = node.value - min_val / (max_val - min_val)  # (the reference's precedence, :52)
        return normalized_value + exploration

    def select_child(self, budget: float, c: float, uav_specifications: Optional[Dict], py_rng: random.Random):
        max_children, max_uct = [], -np.inf
        vals = [child.value for child in self.children]
        lo, hi = min(vals), max(vals)
        for child in self.children:
            uct = self.uct(child, lo, hi, c=c)
            cost = action_costs(child.action, self.action, uav_specifications)
            if cost == 0 or cost >= budget:
                uct = -np.inf
            if max_uct == uct:
                max_children.append(child)
            elif uct > max_uct:
                max_uct, max_children = uct, [child]
        return py_rng.choice(max_children)


class _Search:
    """One root: its tree, its random streams and the generator that walks it."""

    def __init__(self, root_slot: int, previous_action, budget: float, np_seed: int, py_seed: Optional[int]):
        self.root = Node(root_slot, -1, [], parent=None, action=previous_action)
        self.budget = float(budget)
        self.np_rng = np.random.RandomState(np_seed)
        self.py_rng = random.Random(py_seed)
        self.nodes = 1


class ClassicMCTS:
    def __init__(self, engine, cfg, uav_specifications: Optional[Dict], min_altitude: float, max_altitude: float,
                 altitude_spacing: float, num_simulations: int = 100, gamma: float = 0.95, c: float = 2.0,
                 episode_horizon: int = 5, k: float = 4.0, alpha: float = 0.75, epsilon_expand: float = 0.2,
                 epsilon_rollout: float = 0.5, max_greedy_radius: float = 3, use_gcb_rollout: bool = False, adaptive: bool = False,
                 node_capacity: Optional[int] = None):
        """engine: factor-state IPPEngine with window_rows > 0, score_scratch=True and node_capacity > 0 whose env slots hold
        the roots (mean, covariance, ground truth); the other arguments are MCTSMission's (mcts_mission.py:69-165)."""
        self.engine, self.cfg, self.uav = engine, cfg, uav_specifications
        self.num_simulations, self.gamma, self.c = int(num_simulations), float(gamma), float(c)
        self.episode_horizon, self.k, self.alpha = int(episode_horizon), float(k), float(alpha)
        self.epsilon_expand, self.epsilon_rollout = float(epsilon_expand), float(epsilon_rollout)
        self.use_gcb_rollout, self.adaptive = bool(use_gcb_rollout), bool(adaptive)
        self.resolution = float(cfg.resolution)
        self.actions = enumerate_actions(_Grid(cfg), min_altitude, max_altitude, altitude_spacing)
        self.actions_np = action_dict_to_np_array(self.actions)
        self.policy = RolloutPolicy(engine, self.actions_np, uav_specifications, max_greedy_radius, adaptive=True)  # (masked always)
        self.pool = TreeNodePool(engine, int(node_capacity or engine._c.node_capacity))
        if self.episode_horizon > engine.TREE_DEPTH:
            raise ValueError(f"episode_horizon {self.episode_horizon} exceeds the engine's path depth {engine.TREE_DEPTH}")
        if uav_specifications is not None:
            engine.set_uav(uav_specifications["max_v"], uav_specifications["max_a"])
        self.stats = dict(device_steps=0, launches=0, score_calls=0)

    # ------------------------------------------------------------------ policies (generators: `yield` = a device request)
    def _mask(self, node: Node, remaining_budget: float) -> np.ndarray:
        return self.policy.next_actions_mask(node.action, remaining_budget, self.uav)

    def _greedy(self, node: Node, actions: np.ndarray):
        """greedy_action (:232-246): the first maximiser of the predicted reward."""
        rewards = yield ("score", node, actions)
        best, best_r = None, -np.inf
        for a, r in zip(actions, rewards):
            if r > best_r:
                best, best_r = a, r
        return best

    def _eps_greedy(self, s: _Search, node: Node, remaining_budget: float, epsilon: float):
        """:248-256"""
        msk = self._mask(node, remaining_budget)
        available = self.actions_np[msk]
        if s.np_rng.uniform(0, 1) > epsilon and msk.sum() > 0:
            return (yield from self._greedy(node, available))
        return available[s.np_rng.choice(len(available))]

    def _gcb(self, s: _Search, node: Node, remaining_budget: float):
        """gcb_policy (:204-221): softmax of the benefit-to-cost values."""
        available = self.actions_np[self._mask(node, remaining_budget)]
        values = np.asarray((yield ("score", node, available)), dtype=np.float64)
        p = np.exp(values) / np.sum(np.exp(values))
        return available[s.np_rng.choice(len(available), p=p)]

    def _rollout(self, s: _Search, node: Node, remaining_budget: float, previous_action, depth: int):
        """rollout / gcb_rollout (:175-200, :223-239); the nodes it creates are not attached to the tree."""
        if depth == 0 or remaining_budget < self.resolution:
            return 0
        if self.use_gcb_rollout:
            action = yield from self._gcb(s, node, remaining_budget)
        else:
            action = yield from self._eps_greedy(s, node, remaining_budget, self.epsilon_rollout)
        nxt = yield ("step", node, action)
        remaining_budget -= action_costs(action, previous_action, self.uav)
        return nxt.edge_reward + self.gamma * (yield from self._rollout(s, nxt, remaining_budget, node.action, depth - 1))

    def _widen(self, s: _Search, node: Node, remaining_budget: float):
        """progressive_widening (:263-272) -> (next node, expanded?)"""
        n_available = int(self._mask(node, remaining_budget).sum())
        if RolloutPolicy.widen(len(node.children), node.visits, self.k, self.alpha, n_available):
            action = yield from self._eps_greedy(s, node, remaining_budget, self.epsilon_expand)  # expand (:258-261)
            return (yield ("step", node, action)), True
        return node.select_child(remaining_budget, self.c, self.uav, s.py_rng), False

    def _simulate(self, s: _Search, node: Node, remaining_budget: float, depth: int):
        """simulate (:274-304)"""
        if depth == 0 or remaining_budget < self.resolution:
            return 0
        if node.visits == 0:
            value = yield from self._rollout(s, node, remaining_budget, node.action, depth)
            node.visits += 1
            node.value_sum += value
            return value
        nxt, expanded = yield from self._widen(s, node, remaining_budget)
        if expanded:
            node.children.append(nxt)
        remaining_budget -= action_costs(nxt.action, node.action, self.uav)
        value = nxt.edge_reward + (yield from self._simulate(s, nxt, remaining_budget, depth - 1))
        node.visits += 1
        nxt.visits += 1
        node.value_sum += value
        return value

    def _search(self, s: _Search, num_simulations: int):
        for _ in range(num_simulations):
            yield from self._simulate(s, s.root, s.budget, self.episode_horizon)

    # ------------------------------------------------------------------ the driver
    def run(self, searches: Sequence[_Search], num_simulations: Optional[int] = None):
        """Walk every root's search to its end: the pending `step` requests of all roots go into ONE ipp_tree_step launch per
        round, `score` requests are answered one state at a time (ipp_tree_score_actions)."""
        n_sims = self.num_simulations if num_simulations is None else int(num_simulations)
        gens = [self._search(s, n_sims) for s in searches]
        pending, live = [None] * len(gens), []
        for i, g in enumerate(gens):
            try:
                pending[i] = next(g)
                live.append(i)
            except StopIteration:
                pass
        while live:
            answers = {}
            steps = [i for i in live if pending[i][0] == "step"]
            if steps:
                roots = [pending[i][1].root for i in steps]
                parents = [pending[i][1].dev if pending[i][1].dev >= 0 else None for i in steps]
                acts = np.array([pending[i][2] for i in steps], dtype=np.float64)
                prevs = np.array([pending[i][1].action for i in steps], dtype=np.float64)
                reward, status, new = self.pool.expand(roots, parents, acts, prevs, adaptive=True,
                                                       use_flight_time=self.uav is not None)
                if int(status.abs().sum()) != 0:
                    raise ValueError("ipp_tree_step rejected a search step")
                rw = reward.detach().cpu().numpy().astype(np.float64)
                self.stats["device_steps"] += len(steps)
                self.stats["launches"] += 1
                for j, i in enumerate(steps):
                    parent = pending[i][1]
                    child = Node(parent.root, int(new[j]), (parent.path + [int(new[j])]), parent=parent, action=acts[j],
                                 edge_reward=float(rw[j]))
                    searches[i].nodes += 1
                    answers[i] = child
            for i in live:
                if pending[i][0] == "score":
                    node, actions = pending[i][1], pending[i][2]
                    answers[i] = self.policy.score(node.root, node.path, node.action, actions)
                    self.stats["score_calls"] += 1
            still = []
            for i in live:
                try:
                    pending[i] = gens[i].send(answers[i])
                    still.append(i)
                except StopIteration:
                    pass
            live = still
        return [s.root for s in searches]

    def new_search(self, root_slot: int, previous_action, budget: float, worker_id: int = 0, py_seed: Optional[int] = None) -> _Search:
        """A root with the reference's per-worker NumPy seed (run_simulations_proxy, :314)."""
        return _Search(root_slot, previous_action, budget, worker_id * 42 + 1, py_seed)

    @staticmethod
    def merge_roots(root_a: Node, root_b: Node) -> Node:
        """:320-339"""
        # both sides keyed by action: children are not de-duplicated on expansion, so of several children with one action the LAST
        # one stands for the action, and every action of root_b is merged or appended once
        by_action = {str(list(ch.action)): ch for ch in root_a.children}
        by_action_b = {str(list(ch.action)): ch for ch in root_b.children}
        for key, ch_b in by_action_b.items():
            if key in by_action:
                by_action[key].visits += ch_b.visits
                by_action[key].value_sum += ch_b.value_sum
            else:
                root_a.children.append(ch_b)
            root_a.visits += ch_b.visits
            root_a.value_sum += ch_b.value_sum
        return root_a

    @staticmethod
    def select_best_child(root: Node) -> Optional[Node]:
        """:341-351: the FIRST child with the largest value."""
        best = None
        for ch in root.children:
            if best is None or ch.value > best.value:
                best = ch
        return best

    def replan(self, root_slots: Sequence[int], previous_actions, budgets: Sequence[float], py_seeds: Optional[Sequence[int]] = None):
        """replan (:353-389) for many roots: next waypoint per root (one worker each, like the reference's num_workers = 1)."""
        self.pool.clear()
        searches = [self.new_search(int(r), np.asarray(p, dtype=np.float64), float(b), 0, None if py_seeds is None else py_seeds[i])
                    for i, (r, p, b) in enumerate(zip(root_slots, previous_actions, budgets))]
        roots = self.run(searches)
        return [self.select_best_child(r).action for r in roots], roots

    def execute(self, root_slot: int, budget: float, init_action=(2.0, 2.0, 14.0), meas_noise_fn=None, max_steps: int = 1000):
        """The mission loop of one env (:391-416): replan, fly, measure, update, until the budget is spent."""
        prev = np.asarray(init_action, dtype=np.float64)
        remaining, waypoints = float(budget), []
        while remaining >= self.resolution and len(waypoints) < max_steps:
            (wp,), _ = self.replan([root_slot], [prev], [remaining])
            remaining -= action_costs(wp, prev, self.uav)
            eps = None
            if meas_noise_fn is not None:
                eps = np.zeros((1, self.engine.meas_cap))
                e = np.ravel(meas_noise_fn(wp))
                eps[0, : e.size] = e
            self.engine.step(wp[None], prev[None], env_ids=[root_slot], meas_noise=eps, adaptive=self.adaptive,
                             use_flight_time=self.uav is not None)
            waypoints.append(wp)
            prev = wp
        return np.array(waypoints), remaining
