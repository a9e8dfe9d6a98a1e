"""
Batched tree-search driver for the MCTS-zero planner: the reference's AlphaZero-style search
(planning/mcts_zero/mcts.py) for MANY roots at once, with every covariance step of every tree on the GPU.

What the reference does per simulation (mcts.py:166-265): descend from the root; at every expanded node pick the action
with the largest PUCT score (:280-296, forced playouts at the root), take a covariance-only predict step
(simulate_prediction_step) to get the reward and the child's N x N covariance, recurse; at a leaf ask the network for
(policy, value) through a pair of queues, mask the policy with the valid-action mask (:148-158), add Dirichlet noise at
the root of the first simulation (:160-164), back the value up (:255-265).  get_policy (:83-143) turns the root's visit
counts into a policy after pruning forced playouts.

Here:
  * node states never leave the device: a node is the root env slot of its tree plus path-local factor columns
    (ipp_tree_step, csrc/k_tree.h; bookkeeping in planning/tree.py::TreeNodePool).  The predict steps of all roots that
    need one at the same tree level go into ONE ipp_tree_step launch.
  * an edge's masked trace reduction is cached the first time it is computed (the reference recomputes the whole
    N x N update on every traversal): reward = reduction / (cost(previous waypoint, action) + 1) is rebuilt on the host
    from the cached reduction, because the cost depends on the path that led to the node.
  * node identity.  The reference keys nodes by hash(str(P)) (mcts.py:20-21): P printed to 8 digits, so states reached
    by the same measurements in a different order are ONE node (the covariance update commutes), with shared statistics,
    priors and valid-action mask.  Here the key is (root, sorted action indices) -- the same identification without
    printing matrices.  (For maps of more than 31 cells NumPy abbreviates str(P) to the corners of the matrix and almost
    all states of the reference collide; the golden vectors are recorded with the print threshold raised, see
    tests/golden/gen_golden.py.)
  * per-node statistics are stored on the node's VALID actions only (the reference keeps num_actions-long arrays and
    sets the score of invalid ones to -inf); the zero entries of the invalid actions still take part in the min-max
    normalisation of Q (mcts.py:267-278), as they do there.
  * the network is a callable `infer(requests) -> [(policy or None, value), ...]`; leaf evaluation stays stock PyTorch
    (out of scope).  The reference builds network inputs through
    generate_input_feature_planes, which zeroes rows / columns of the live node states in place when adaptive
    (features.py:98-99); that side effect is NOT reproduced.
  * `sims_in_flight` > 1 runs that many simulations per root between backups (virtual visits keep them apart), so that a
    launch carries roots x sims_in_flight items; 1 reproduces the reference's sequential search exactly (golden test).
  * randomness: every root draws from its own generator; with `rng=np.random` (one root) the call sequence is the
    reference's (np.random.choice per PUCT argmax, np.random.dirichlet at the root).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..common.actions import action_costs, action_dict_to_np_array, enumerate_actions
from ..tree import TreeNodePool


class _Grid:
    def __init__(self, cfg):
        self.x_dim, self.y_dim, self.resolution = cfg.x_dim, cfg.y_dim, cfg.resolution
        self.num_grid_cells = cfg.n_cells


class _Node:
    __slots__ = ("key", "root", "dev", "dev_path", "parent", "parent_k", "expanded", "idx", "Ps", "Nsa", "Qsa", "child", "num",
                 "Ns", "pending_value")

    def __init__(self, key, root):
        self.key, self.root = key, root
        self.dev = -1            # device node id (TreeNodePool), -1: not stored (the root env slot itself has dev_path [])
        self.dev_path: Optional[List[int]] = None
        self.parent = None       # (node, k) the device state is (or will be) derived from
        self.expanded = False    # has priors (the reference's `node_rep in self.Ps`)
        self.idx = None          # valid action indices, ascending (Vs)
        self.Ps = self.Nsa = self.Qsa = self.child = self.num = None
        self.Ns = 0
        self.pending_value = None


class BatchedMCTS:
    def __init__(self, engine, hyper_params: Dict, meta_data: Dict, infer: Callable, node_capacity: Optional[int] = None,
                 sims_in_flight: int = 1, tie_break: str = "reference", row_costs: bool = False):
        """engine: factor-state IPPEngine with window_rows > 0 and node_capacity > 0 whose env slots hold the roots.
        hyper_params / meta_data: the reference's dictionaries (mcts.py:25-49).
        tie_break: "reference" = np.random.choice among equal PUCT scores (mcts.py:237); "first" = the lowest action index
        (deterministic; what VectorMCTS is compared with).  row_costs: travel costs by the array formula of VectorMCTS
        (same value up to the last bit of a 3-term sum) instead of actions.action_costs."""
        self.tie_break, self.row_costs = tie_break, row_costs
        self.engine = engine
        self.hp, self.meta = hyper_params, meta_data
        self.infer = infer
        self.pool = TreeNodePool(engine, node_capacity or engine._c.node_capacity)
        self.uav = meta_data["uav_specifications"]
        self.horizon = int(meta_data["episode_horizon"])
        self.gamma = float(hyper_params["gamma"])
        self.puct_init, self.puct_base = float(hyper_params["puct_init"]), float(hyper_params["puct_base"])
        self.fpf = float(hyper_params["forced_playout_factor"])
        self.alpha, self.eps = float(hyper_params["dirichlet_alpha"]), float(hyper_params["dirichlet_eps"])
        self.max_dist = float(hyper_params["max_valid_action_distance"])
        self.num_simulations = int(hyper_params["num_mcts_simulations"])
        self.adaptive = meta_data.get("scenario_info") is not None
        if self.adaptive:
            engine.set_adaptive(meta_data["scenario_info"]["value_threshold"], meta_data["scenario_info"]["interval_factor"])
        if self.uav is not None:
            engine.set_uav(self.uav["max_v"], self.uav["max_a"])
        self.actions_np = action_dict_to_np_array(enumerate_actions(
            _Grid(engine.cfg), meta_data["min_altitude"], meta_data["max_altitude"], meta_data["altitude_spacing"]))
        self.sims_in_flight = max(1, int(sims_in_flight))
        if self.horizon + 1 > engine.TREE_DEPTH:
            raise ValueError(f"episode_horizon {self.horizon} needs paths of {self.horizon + 1} steps; the engine holds {engine.TREE_DEPTH}")
        self.stats = dict(device_steps=0, launches=0, inferences=0, revisits=0, new_visits=0, nodes=0)
        self._cell_index = None

    @property
    def num_actions(self) -> int:
        return self.actions_np.shape[0]

    # ------------------------------------------------------------------ reference formulas
    def next_actions_mask(self, position, budget) -> np.ndarray:
        """mcts.py:148-158 as simulate() calls it (no UAV argument: distance-based)."""
        d = np.linalg.norm(self.actions_np - position, ord=2, axis=1)
        return (d > 0) & (d <= budget) & (d < self.max_dist)

    DENSE_ACTIONS = 4096  # up to this many actions the reference's num_actions-long arrays are used as they are

    def _valid_idx(self, position, budget):
        """Ascending indices of the valid actions and (small action sets only) the dense mask.  Large maps: only the
        actions in the cells within max_valid_action_distance of the position are looked at (the reference evaluates the
        distance to all num_actions actions for every leaf: 80 000 at 200x200 with two altitude levels)."""
        if self.num_actions <= self.DENSE_ACTIONS:
            mask = self.next_actions_mask(position, budget)
            return np.nonzero(mask)[0], mask
        if self._cell_index is None:
            res = self.engine.cfg.resolution
            cx = np.floor(self.actions_np[:, 0] / res).astype(np.int64)
            cy = np.floor(self.actions_np[:, 1] / res).astype(np.int64)
            W, H = int(cx.max()) + 1, int(cy.max()) + 1
            order = np.argsort(cy * W + cx, kind="stable")
            counts = np.bincount((cy * W + cx)[order], minlength=W * H)
            self._cell_index = (W, H, order, np.concatenate([[0], np.cumsum(counts)]))
        W, H, order, start = self._cell_index
        res = self.engine.cfg.resolution
        k = int(np.ceil(self.max_dist / res)) + 1
        px, py = int(np.floor(position[0] / res)), int(np.floor(position[1] / res))
        x0, x1 = max(px - k, 0), min(px + k, W - 1)
        rows = range(max(py - k, 0), min(py + k, H - 1) + 1)
        if x0 > x1 or len(rows) == 0:
            return np.zeros(0, dtype=np.int64), None
        cand = np.concatenate([order[start[y * W + x0]:start[y * W + x1 + 1]] for y in rows])  # cells of a row are adjacent
        d = np.linalg.norm(self.actions_np[cand] - position, ord=2, axis=1)
        return np.sort(cand[(d > 0) & (d <= budget) & (d < self.max_dist)]), None

    def row_cost(self, actions: np.ndarray, prevs: np.ndarray) -> np.ndarray:
        """actions.py:8-41 for rows of (action, previous waypoint) pairs."""
        diff = actions - prevs
        dist = np.sqrt((diff * diff).sum(axis=1))
        if self.uav is None:
            return dist
        v, a = float(self.uav["max_v"]), float(self.uav["max_a"])
        ramp = np.minimum(0.5 * dist, np.square(v) / (2 * a))
        return (dist - 2 * ramp) / v + 2 * np.sqrt(2 * ramp / a)

    @staticmethod
    def _normalize_q(q: np.ndarray, has_outside: bool) -> np.ndarray:
        """mcts.py:267-278 over the num_actions-long array whose entries outside the valid set are 0."""
        if np.all(q == 0):
            return q
        lo, hi = float(q.min()), float(q.max())
        if has_outside:
            lo, hi = min(lo, 0.0), max(hi, 0.0)
        if lo == hi:
            return q / hi
        return (q - lo) / (hi - lo)

    def _uct(self, nd: _Node, force_playouts: bool, nsa=None) -> np.ndarray:
        """mcts.py:280-296 on the valid actions (the invalid ones are -inf there)."""
        nsa = nd.Nsa if nsa is None else nsa
        qn = self._normalize_q(nd.Qsa, len(nd.idx) < self.num_actions)
        prior = self.puct_init + np.log((nd.Ns + self.puct_base + 1) / self.puct_base)
        prior = prior * (nd.Ps * (np.sqrt(nd.Ns + 1) / (1 + nsa)))
        uct = qn + prior
        if force_playouts:
            nfp = np.ceil(np.sqrt(self.fpf * nd.Ps * nd.Ns))
            nfp[nsa == 0] = 0
            uct = np.where(nsa < nfp, np.inf, uct)
        return uct

    # ------------------------------------------------------------------ search
    def get_policy(self, roots: Sequence[int], previous_actions, budgets, depth: int = 0, temperature: float = 1.0,
                   deploy_time: bool = False, rngs=None):
        """The reference's get_policy for every root env slot in `roots`: returns a list of (policy, second) tuples --
        second = valid-action mask of the root (temperature > 0) or the pruned visit counts (temperature == 0) -- or None
        where no valid action was visited (mcts.py:127-129).  rngs: one generator per root (np.random-like: choice,
        dirichlet); default RandomState(root id)."""
        R = len(roots)
        prev0 = np.asarray(previous_actions, dtype=np.float64).reshape(R, 3)
        budget0 = np.asarray(budgets, dtype=np.float64).reshape(R)
        rngs = list(rngs) if rngs is not None else [np.random.RandomState(int(r)) for r in roots]
        self.pool.clear()  # reset_mcts_each_step: a fresh tree per call
        ids: List[Dict[Tuple, _Node]] = [dict() for _ in range(R)]
        root_nodes = []
        for j, r in enumerate(roots):
            nd = _Node((), j)
            nd.dev_path = []
            ids[j][()] = nd
            root_nodes.append(nd)
        sim = 0
        while sim < self.num_simulations:
            w = min(self.sims_in_flight, self.num_simulations - sim)
            self._wave(roots, ids, root_nodes, prev0, budget0, depth, rngs, sim, w)
            sim += w
        self.stats["nodes"] = sum(sum(1 for n in d.values() if n.expanded) for d in ids)
        self.last_roots = root_nodes  # statistics of the finished search (tests, callers that want Q / N of the root)
        return [self._policy_from_root(root_nodes[j], prev0[j], budget0[j], temperature, deploy_time, rngs[j]) for j in range(R)]

    # one wave = `w` simulations per root: host descents -> batched device steps per level -> inference -> backups
    def _wave(self, roots, ids, root_nodes, prev0, budget0, depth0, rngs, sim0, w):
        R = len(roots)
        sims = []  # per simulation: dict(j, path [(node, k, cost)], terminal, leaf info)
        requests_by_level: Dict[int, List] = {}
        for s in range(w):
            for j in range(R):
                sims.append(self._descend(j, ids[j], root_nodes[j], prev0[j], float(budget0[j]), depth0, rngs[j], sim0 + s,
                                          requests_by_level, virtual=(w > 1)))
        # ---- device: predict / expand steps level by level (a child is derived from its parent's stored state)
        for level in sorted(requests_by_level):
            self._device_level(roots, requests_by_level[level])
        # ---- inference for the leaves reached in this wave (one request per leaf node)
        leaves = [s for s in sims if s["terminal"] == "leaf"]
        todo, seen = [], {}
        for s in leaves:
            nd = s["leaf"]
            if id(nd) not in seen:
                seen[id(nd)] = len(todo)
                todo.append(s)
        if todo:
            # action_msk: the reference's dense mask (None on large maps, where valid_idx lists the valid actions)
            replies = self.infer([dict(root=int(roots[s["j"]]), path=list(s["leaf"].key), action_msk=s["mask"], valid_idx=s["idx"],
                                       depth=s["depth"], previous_action=s["prev"], budget=s["budget"]) for s in todo])
            self.stats["inferences"] += len(todo)
            for s, (policy, value) in zip(todo, replies):
                self._expand_leaf(s["leaf"], s["idx"], s["mask"], policy, float(value), s["depth"] == 0 and s["sim"] == 0, rngs[s["j"]])
        # ---- backups, in simulation order
        for s in sims:
            if s["terminal"] == "leaf":
                value = s["leaf"].pending_value
            else:
                value = 0.0
            for nd, k, cost in reversed(s["path"]):
                if s["virtual"]:
                    nd.Nsa[k] -= 1
                    nd.Ns -= 1
                reward = nd.num[k] / (cost + 1.0)  # rewards.py:31
                value = reward + self.gamma * value
                if nd.Nsa[k] > 0:
                    nd.Qsa[k] = (nd.Nsa[k] * nd.Qsa[k] + value) / (nd.Nsa[k] + 1)
                    nd.Nsa[k] += 1
                    self.stats["revisits"] += 1
                else:
                    nd.Qsa[k] = value
                    nd.Nsa[k] = 1
                    self.stats["new_visits"] += 1
                nd.Ns += 1

    def _descend(self, j, table, node, prev, budget, depth, rng, sim, requests_by_level, virtual):
        path = []
        out = dict(j=j, sim=sim, path=path, terminal="zero", virtual=virtual)
        prev = np.array(prev, dtype=np.float64)
        while True:
            if depth > self.horizon or budget <= 0:  # mcts.py:175-176
                return out
            if not node.expanded:
                idx, mask = self._valid_idx(prev, budget)
                if len(idx) == 0:  # mcts.py:201-202
                    return out
                out.update(terminal="leaf", leaf=node, idx=idx, mask=mask, depth=depth, prev=prev.copy(), budget=budget)
                return out
            uct = self._uct(node, force_playouts=(depth == 0))
            ties = (uct == np.max(uct)).nonzero()[0]
            k = int(ties[0]) if self.tie_break == "first" else int(rng.choice(ties))  # mcts.py:237
            a_idx = int(node.idx[k])
            action = self.actions_np[a_idx]
            cost = float(self.row_cost(action[None], prev[None])[0]) if self.row_costs else float(action_costs(action, prev, self.uav))
            child = node.child[k]
            if child is None:
                key = tuple(sorted(node.key + (a_idx,)))
                child = table.get(key)
                if child is None:
                    child = _Node(key, j)
                    child.parent = (node, k)
                    table[key] = child
                node.child[k] = child
            if np.isnan(node.num[k]):  # first traversal of this edge: one device step
                node.num[k] = np.inf  # (requested; filled by _device_level)
                store = depth + 1 <= self.horizon and child.dev < 0 and child.dev_path is None and child.parent == (node, k)
                requests_by_level.setdefault(depth, []).append((j, node, k, a_idx, prev.copy(), cost, child if store else None))
            path.append((node, k, cost))
            if virtual:
                node.Nsa[k] += 1
                node.Ns += 1
            budget -= cost
            prev = action.copy()
            node = child
            depth += 1

    def _device_level(self, roots, reqs):
        """One ipp_tree_step launch: item = (root slot, parent path, action, previous waypoint[, new node id])."""
        n = len(reqs)
        root_ids = np.array([int(roots[j]) for j, *_ in reqs], dtype=np.int32)
        acts = np.stack([self.actions_np[a_idx] for _, _, _, a_idx, _, _, _ in reqs])
        prevs = np.stack([p for _, _, _, _, p, _, _ in reqs])
        D = self.engine.TREE_DEPTH
        paths = np.full((n, D), -1, dtype=np.int32)
        new_ids = np.full(n, -1, dtype=np.int32)
        for i, (j, nd, k, a_idx, p, cost, child) in enumerate(reqs):
            dp = nd.dev_path
            if dp is None:
                raise RuntimeError("a node below an unstored node was asked for a step (tree bookkeeping error)")
            paths[i, : len(dp)] = dp
            if child is not None:
                new_ids[i] = self.pool.allocate()
        reward, status = self.engine.tree_step(root_ids, paths, acts, prevs, new_ids=new_ids, adaptive=self.adaptive,
                                               use_flight_time=self.uav is not None)
        r = reward.detach().cpu().numpy().astype(np.float64)
        st = status.detach().cpu().numpy()
        if np.any(st != 0):
            raise RuntimeError(f"ipp_tree_step reported status {st[st != 0][:4]} (rank_cap / footprint)")
        self.stats["device_steps"] += n
        self.stats["launches"] += 1
        for i, (j, nd, k, a_idx, p, cost, child) in enumerate(reqs):
            nd.num[k] = r[i] * (cost + 1.0)  # masked trace reduction of the edge (the kernel divided by cost + 1)
            if child is not None:
                child.dev = int(new_ids[i])
                child.dev_path = nd.dev_path + [child.dev]

    def _expand_leaf(self, nd: _Node, idx, mask, policy, value, noise: bool, rng):
        """mcts.py:204-233: priors masked, Dirichlet noise at the root of the first simulation, normalised."""
        A = self.num_actions
        dense = noise or policy is not None or mask is not None  # small action sets: the reference's array arithmetic to the last bit
        if dense:
            if mask is None:
                mask = np.zeros(A, dtype=bool)
                mask[idx] = True
            full = (np.ones(A) / A if policy is None else np.asarray(policy, dtype=np.float64)) * mask
            if noise:  # add_exploration_noise works on the num_actions-long vector (invalid actions receive noise too)
                full = (1 - self.eps) * full + self.eps * rng.dirichlet([self.alpha] * A)
                full = full / np.sum(full)
            total = np.sum(full)
            ps = full[idx]
        else:
            ps = np.full(len(idx), 1.0 / A)
            total = float(np.sum(ps))
        if total > 0:
            ps = ps / total
        else:  # "All valid moves have 0 probability" (mcts.py:226-229)
            ps = np.full(len(idx), 1.0 / len(idx))
        nd.idx, nd.Ps = idx, ps
        nd.Nsa, nd.Qsa = np.zeros(len(idx)), np.zeros(len(idx))
        nd.child = [None] * len(idx)
        nd.num = np.full(len(idx), np.nan)
        nd.Ns = 0
        nd.expanded = True
        nd.pending_value = value

    def _policy_from_root(self, root: _Node, prev, budget, temperature, deploy_time, rng):
        """mcts.py:98-143."""
        A = self.num_actions
        visits = np.zeros(A)
        if root.expanded:
            visits[root.idx] = root.Nsa
        if not deploy_time and root.expanded:
            nsa = root.Nsa
            best = int(rng.choice((visits == np.max(visits)).nonzero()[0]))
            nfp = np.zeros(A)
            nfp[root.idx] = np.ceil(np.sqrt(self.fpf * root.Ps * root.Ns))
            nfp[visits == 0] = 0
            best_k = int(np.nonzero(root.idx == best)[0][0]) if best in root.idx else None
            uct = self._uct(root, force_playouts=False)
            max_puct = uct[best_k] if best_k is not None else -np.inf
            qn = self._normalize_q(root.Qsa, len(root.idx) < A)
            for k, a_idx in enumerate(root.idx):
                if a_idx == best or nfp[a_idx] <= 0:
                    continue
                for _ in range(int(nfp[a_idx])):
                    visits[a_idx] -= 1
                    prior = self.puct_init + np.log((root.Ns + self.puct_base + 1) / self.puct_base)
                    prior *= root.Ps[k] * (np.sqrt(root.Ns + 1) / (1 + visits[a_idx]))
                    if qn[k] + prior >= max_puct:
                        visits[a_idx] += 1
                        break
            visits[visits == 1] = 0
        if np.sum(visits) == 0:
            return None
        if temperature == 0:
            best = int(rng.choice(np.array(np.argwhere(visits == np.max(visits))).flatten()))
            policy = [0] * A
            policy[best] = 1
            return policy, visits
        vt = visits ** (1.0 / temperature)  # (the reference loops over the actions: same IEEE pow per element)
        return (vt / np.sum(vt)).tolist(), self.next_actions_mask(prev, budget)
