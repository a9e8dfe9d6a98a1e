"""
VectorMCTS: the tree search of mcts.py (BatchedMCTS) with the host side vectorised over the ROOTS.

BatchedMCTS walks every root's simulation in Python (about 0.2 ms of NumPy per simulation): right for one root and for
the golden comparison with the reference, hopeless for BASELINE configs[4] (1024 roots x 256 simulations = 2.6e5
descents per search).  Here one NumPy operation advances the current simulation of ALL roots by one tree level: node
statistics live in padded tables [nodes, Kmax] (Kmax = the most valid actions any position can have), PUCT scores, the
arg-max, costs, budgets and backups are array operations over the roots, the valid-action sets of a whole wave of
leaves are computed at once from the grid geometry, and only the transposition lookup (one dict operation per
descent step) and the device requests stay per item.  Device work is unchanged: one ipp_tree_step launch per tree level
and wave for all roots (csrc/k_tree.h).

Same search as BatchedMCTS(sims_in_flight=W): same formulas (mcts.py:83-296 of the reference), same node identity
(states reached by the same measurements in any order are one node; the key is a commutative 64-bit hash of the action
multiset instead of a sorted tuple), same order of simulations and backups.  With tie_break="first" (lowest action index
among equal scores) both drivers build identical trees (tests/test_hip_mcts.py); "random" breaks ties with one uniform
draw per candidate from a single generator, which cannot reproduce the reference's per-root np.random.choice stream.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .mcts import BatchedMCTS


class VectorMCTS(BatchedMCTS):
    def __init__(self, engine, hyper_params: Dict, meta_data: Dict, infer: Callable, node_capacity: Optional[int] = None,
                 sims_in_flight: int = 4, tie_break: str = "random", seed: int = 0):
        super().__init__(engine, hyper_params, meta_data, infer, node_capacity, sims_in_flight)
        if tie_break not in ("first", "random"):
            raise ValueError("tie_break must be 'first' or 'random'")
        self.tie_break = tie_break
        self.rng = np.random.RandomState(seed)
        self._uniform: Dict[int, float] = {}
        cfg = engine.cfg
        res = cfg.resolution
        self._res = res
        self._W, self._H = cfg.x_dim, cfg.y_dim
        # candidate offsets around a position: every cell within ceil(max_dist / res) + 1 cells, every altitude level
        self._levels = np.unique(self.actions_np[:, 2])
        self._n_lv = len(self._levels)
        self._ncell = cfg.x_dim * cfg.y_dim
        # action index of (level, col, row) in the reference's enumeration (actions.py:73-82: idx = level N + x_dim col + row)
        cx = np.floor(self.actions_np[: self._ncell, 0] / res).astype(np.int64)
        cy = np.floor(self.actions_np[: self._ncell, 1] / res).astype(np.int64)
        self._cell_action = np.full((self._W, self._H), -1, dtype=np.int64)
        self._cell_action[cx, cy] = np.arange(self._ncell)
        k = int(np.ceil(self.max_dist / res)) + 1
        d = np.arange(-k, k + 1)
        dx, dy = np.meshgrid(d, d, indexing="ij")
        self._off_x, self._off_y = dx.ravel(), dy.ravel()
        self.Kmax = len(self._off_x) * self._n_lv
        self._z = np.random.RandomState(12345).randint(1, 2 ** 62, size=self.num_actions, dtype=np.int64).astype(np.uint64)

    # ------------------------------------------------------------------ valid-action sets of many positions at once
    def valid_sets(self, pos: np.ndarray, budget: np.ndarray):
        """mcts.py:148-158 for n positions: (idx [n, Kmax] ascending, -1 padded; K [n])."""
        n = len(pos)
        px = np.floor(pos[:, 0] / self._res).astype(np.int64)
        py = np.floor(pos[:, 1] / self._res).astype(np.int64)
        cx = px[:, None] + self._off_x[None, :]
        cy = py[:, None] + self._off_y[None, :]
        inside = (cx >= 0) & (cx < self._W) & (cy >= 0) & (cy < self._H)
        base = self._cell_action[np.clip(cx, 0, self._W - 1), np.clip(cy, 0, self._H - 1)]  # [n, n_off] level-0 action of the cell
        idx = (base[:, None, :] + (np.arange(self._n_lv) * self._ncell)[None, :, None]).reshape(n, -1)  # [n, levels * n_off]
        ok = np.broadcast_to(inside[:, None, :], (n, self._n_lv, inside.shape[1])).reshape(n, -1)
        diff = self.actions_np[idx] - pos[:, None, :]
        dist = np.sqrt((diff * diff).sum(axis=2))  # == np.linalg.norm(.., ord=2, axis=1) of the reference, row by row
        ok = ok & (dist > 0) & (dist <= budget[:, None]) & (dist < self.max_dist)
        big = np.iinfo(np.int64).max
        srt = np.sort(np.where(ok, idx, big), axis=1)
        K = ok.sum(axis=1)
        srt[srt == big] = -1
        return srt[:, : self.Kmax], K

    # ------------------------------------------------------------------ tables
    def _alloc_tables(self, cap):
        K = self.Kmax
        self.t_idx = np.full((cap, K), -1, dtype=np.int64)
        self.t_Ps = np.zeros((cap, K))
        self.t_Nsa = np.zeros((cap, K))
        self.t_Qsa = np.zeros((cap, K))
        self.t_num = np.full((cap, K), np.nan)
        self.t_child = np.full((cap, K), -1, dtype=np.int64)
        self.n_K = np.zeros(cap, dtype=np.int64)
        self.n_Ns = np.zeros(cap)
        self.n_expanded = np.zeros(cap, dtype=bool)
        self.n_hash = np.zeros(cap, dtype=np.uint64)
        self.n_depth = np.zeros(cap, dtype=np.int64)
        self.n_dev = np.full(cap, -1, dtype=np.int64)
        self.n_devpath = np.full((cap, self.engine.TREE_DEPTH), -1, dtype=np.int32)
        self.n_stored = np.zeros(cap, dtype=bool)   # has (or is about to get) a device node / is a root
        self.n_value = np.zeros(cap)                # value returned by the network when the node was expanded
        self.n_count = 0

    def _new_nodes(self, n):
        first = self.n_count
        self.n_count += n
        if self.n_count > len(self.n_K):
            raise RuntimeError("node table exhausted")
        return np.arange(first, first + n)

    def _uct_rows(self, nodes, force_playouts: bool):
        """mcts.py:280-296 for the given nodes: [n, Kmax], -inf on the padding."""
        valid = self.t_idx[nodes] >= 0
        q = self.t_Qsa[nodes]
        nsa = self.t_Nsa[nodes]
        ps = self.t_Ps[nodes]
        ns = self.n_Ns[nodes][:, None]
        qm = np.where(valid, q, 0.0)
        has_outside = (self.n_K[nodes] < self.num_actions)[:, None]
        lo = np.where(valid, q, np.inf).min(axis=1, keepdims=True)
        hi = np.where(valid, q, -np.inf).max(axis=1, keepdims=True)
        lo = np.where(has_outside, np.minimum(lo, 0.0), lo)
        hi = np.where(has_outside, np.maximum(hi, 0.0), hi)
        allzero = np.all(qm == 0, axis=1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            qn = np.where(allzero, qm, np.where(lo == hi, qm / hi, (qm - lo) / (hi - lo)))
        prior = self.puct_init + np.log((ns + self.puct_base + 1) / self.puct_base)
        prior = prior * (ps * (np.sqrt(ns + 1) / (1 + nsa)))
        uct = qn + prior
        if force_playouts:
            nfp = np.ceil(np.sqrt(self.fpf * ps * ns))
            nfp[nsa == 0] = 0
            uct = np.where(nsa < nfp, np.inf, uct)
        return np.where(valid, uct, -np.inf)

    # ------------------------------------------------------------------ search
    def get_policy(self, roots: Sequence[int], previous_actions, budgets, depth: int = 0, temperature: float = 1.0,
                   deploy_time: bool = False, rngs=None):
        """Like BatchedMCTS.get_policy.  rngs (one np.random-like generator per root) are used for the Dirichlet noise of
        small action sets and for get_policy's own draws; default RandomState(root id)."""
        R = len(roots)
        roots_np = np.asarray(roots, dtype=np.int32)
        prev0 = np.asarray(previous_actions, dtype=np.float64).reshape(R, 3)
        budget0 = np.asarray(budgets, dtype=np.float64).reshape(R)
        rngs = list(rngs) if rngs is not None else [np.random.RandomState(int(r)) for r in roots]
        self.pool.clear()
        self._alloc_tables(R * (self.num_simulations + 2) + 8)
        root_ids = self._new_nodes(R)
        self.n_stored[root_ids] = True
        self.n_hash[root_ids] = (np.arange(R, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        self._tables: List[Dict[int, int]] = [dict() for _ in range(R)]
        for j in range(R):
            self._tables[j][int(self.n_hash[root_ids[j]])] = int(root_ids[j])
        sim = 0
        while sim < self.num_simulations:
            w = min(self.sims_in_flight, self.num_simulations - sim)
            self._vwave(roots_np, root_ids, prev0, budget0, depth, rngs, sim, w)
            sim += w
        self.stats["nodes"] = int(self.n_expanded[: self.n_count].sum())
        self.root_ids = root_ids
        return [self._policy_sparse(int(root_ids[j]), prev0[j], float(budget0[j]), temperature, deploy_time, rngs[j]) for j in range(R)]

    def _vwave(self, roots_np, root_ids, prev0, budget0, depth0, rngs, sim0, W):
        R = len(roots_np)
        D = self.horizon + 1 - depth0  # most levels a descent can take
        virtual = W > 1
        p_node = np.full((W, R, max(D, 1)), -1, dtype=np.int64)
        p_k = np.zeros((W, R, max(D, 1)), dtype=np.int64)
        p_cost = np.zeros((W, R, max(D, 1)))
        p_len = np.zeros((W, R), dtype=np.int64)
        leaf = np.full((W, R), -1, dtype=np.int64)
        requests: Dict[int, List] = {}
        pending: Dict[int, tuple] = {}  # leaf node -> (idx row, K, depth, sim, j) of the FIRST simulation that reached it
        ar = np.arange(R)
        for w in range(W):
            cur = root_ids.copy()
            prev = prev0.copy()
            budget = budget0.copy()
            active = np.ones(R, dtype=bool)
            for d in range(depth0, self.horizon + 1):
                active &= budget > 0  # mcts.py:175-176
                if not active.any():
                    break
                exp = self.n_expanded[cur]
                lf = active & ~exp
                if lf.any():
                    rows = np.nonzero(lf)[0]
                    idx, K = self.valid_sets(prev[rows], budget[rows])
                    for i, j in enumerate(rows):
                        if K[i] == 0:  # mcts.py:201-202: returns 0 and stays a leaf
                            continue
                        nd = int(cur[j])
                        leaf[w, j] = nd
                        if nd not in pending:
                            pending[nd] = (idx[i], int(K[i]), d, sim0 + w, int(j), prev[j].copy(), float(budget[j]))
                    active &= ~lf
                sel = active & exp
                if not sel.any():
                    break
                rows = np.nonzero(sel)[0]
                nodes = cur[rows]
                uct = self._uct_rows(nodes, force_playouts=(d == 0))
                top = uct == uct.max(axis=1, keepdims=True)
                if self.tie_break == "first":
                    k = np.argmax(top, axis=1)
                else:
                    k = np.argmax(np.where(top, self.rng.random_sample(top.shape), -1.0), axis=1)
                a_idx = self.t_idx[nodes, k]
                action = self.actions_np[a_idx]
                cost = self.row_cost(action, prev[rows])  # actions.py:8-41
                child = self.t_child[nodes, k]
                need = np.nonzero(child < 0)[0]
                if len(need):
                    hashes = self.n_hash[nodes[need]] + self._z[a_idx[need]]  # commutative: order of the measurements is irrelevant
                    for i, hsh in zip(need, hashes):
                        j = int(rows[i])
                        tab = self._tables[j]
                        c = tab.get(int(hsh))
                        if c is None:
                            c = int(self._new_nodes(1)[0])
                            tab[int(hsh)] = c
                            self.n_hash[c] = hsh
                            self.n_depth[c] = d + 1
                        child[i] = c
                    self.t_child[nodes[need], k[need]] = child[need]
                fresh = np.nonzero(np.isnan(self.t_num[nodes, k]))[0]  # first traversal of the edge: one device step
                if len(fresh):
                    self.t_num[nodes[fresh], k[fresh]] = np.inf
                    for i in fresh:
                        c = int(child[i])
                        store = d + 1 <= self.horizon and not self.n_stored[c]
                        if store:
                            self.n_stored[c] = True
                        requests.setdefault(d, []).append((int(rows[i]), int(nodes[i]), int(k[i]), int(a_idx[i]), prev[rows[i]].copy(),
                                                           float(cost[i]), c if store else -1))
                step = p_len[w, rows]
                p_node[w, rows, step] = nodes
                p_k[w, rows, step] = k
                p_cost[w, rows, step] = cost
                p_len[w, rows] = step + 1
                if virtual:
                    self.t_Nsa[nodes, k] += 1
                    self.n_Ns[nodes] += 1
                budget[rows] -= cost
                prev[rows] = action
                cur[rows] = child
        # ---- device: level by level
        for level in sorted(requests):
            self._vdevice(roots_np, requests[level])
        # ---- inference + expansion of the leaves of this wave
        if pending:
            nodes = list(pending)
            infos = [pending[nd] for nd in nodes]
            dense = self.num_actions <= self.DENSE_ACTIONS
            reqs = []
            for nd, (idx, K, d, s, j, pv, bg) in zip(nodes, infos):
                mask = None
                if dense:
                    mask = np.zeros(self.num_actions, dtype=bool)
                    mask[idx[:K]] = True
                reqs.append(dict(root=int(roots_np[j]), node=nd, action_msk=mask, valid_idx=idx[:K], depth=d, previous_action=pv, budget=bg))
            replies = self.infer(reqs)
            self.stats["inferences"] += len(reqs)
            plain = []  # uniform priors, no noise, sparse arithmetic: all such leaves are written into the tables at once
            for nd, (idx, K, d, s, j, pv, bg), rq, (policy, value) in zip(nodes, infos, reqs, replies):
                if policy is None and rq["action_msk"] is None and not (d == 0 and s == 0):
                    plain.append((nd, idx, K, float(value)))
                else:
                    self._vexpand(nd, idx, K, rq["action_msk"], policy, float(value), d == 0 and s == 0, rngs[j])
            if plain:
                nds = np.array([p[0] for p in plain])
                idxs = np.stack([p[1] for p in plain])
                Ks = np.array([p[2] for p in plain])
                # Ps = (1/A) mask / sum((1/A) mask): K equal entries (mcts.py:204,222-225)
                x = 1.0 / self.num_actions
                for K in np.unique(Ks):  # (the same floating-point sum as the one-leaf path, per distinct K)
                    if int(K) not in self._uniform:
                        self._uniform[int(K)] = x / float(np.sum(np.full(int(K), x)))
                each = np.array([self._uniform[int(K)] for K in Ks])
                self.t_idx[nds] = idxs
                self.t_Ps[nds] = np.where(idxs >= 0, each[:, None], 0.0)
                self.n_K[nds] = Ks
                self.n_Ns[nds] = 0
                self.n_expanded[nds] = True
                self.n_value[nds] = np.array([p[3] for p in plain])
        # ---- backups, simulation by simulation (all roots at once)
        for w in range(W):
            value = np.where(leaf[w] >= 0, self.n_value[np.maximum(leaf[w], 0)], 0.0)
            for stp in range(p_node.shape[2] - 1, -1, -1):
                rows = np.nonzero(p_len[w] > stp)[0]
                if not len(rows):
                    continue
                nodes, k, cost = p_node[w, rows, stp], p_k[w, rows, stp], p_cost[w, rows, stp]
                if virtual:
                    self.t_Nsa[nodes, k] -= 1
                    self.n_Ns[nodes] -= 1
                reward = self.t_num[nodes, k] / (cost + 1.0)  # rewards.py:31
                val = reward + self.gamma * value[rows]
                nsa = self.t_Nsa[nodes, k]
                seen = nsa > 0
                self.t_Qsa[nodes, k] = np.where(seen, (nsa * self.t_Qsa[nodes, k] + val) / (nsa + 1), val)
                self.t_Nsa[nodes, k] = nsa + 1
                self.n_Ns[nodes] += 1
                self.stats["revisits"] += int(seen.sum())
                self.stats["new_visits"] += int((~seen).sum())
                value[rows] = val

    def _vdevice(self, roots_np, reqs):
        n = len(reqs)
        js = np.array([r[0] for r in reqs])
        parents = np.array([r[1] for r in reqs])
        ks = np.array([r[2] for r in reqs])
        acts = self.actions_np[np.array([r[3] for r in reqs])]
        prevs = np.stack([r[4] for r in reqs])
        costs = np.array([r[5] for r in reqs])
        kids = np.array([r[6] for r in reqs])
        paths = self.n_devpath[parents].copy()
        new_ids = np.full(n, -1, dtype=np.int32)
        st = kids >= 0
        if st.any():
            ids = self.pool.allocate(int(st.sum()))
            new_ids[st] = np.atleast_1d(ids)
        rs, sts = [], []
        step = int(getattr(self.engine, "max_batch", n) or n)
        for lo in range(0, n, step):  # (a level of many simulations in flight can exceed the engine's launch size)
            hi = min(n, lo + step)
            reward, status = self.engine.tree_step(roots_np[js[lo:hi]], paths[lo:hi], acts[lo:hi], prevs[lo:hi], new_ids=new_ids[lo:hi],
                                                   adaptive=self.adaptive, use_flight_time=self.uav is not None)
            rs.append(reward.detach().cpu().numpy().astype(np.float64))
            sts.append(status.detach().cpu().numpy())
        r, stt = np.concatenate(rs), np.concatenate(sts)
        if np.any(stt != 0):
            raise RuntimeError(f"ipp_tree_step reported status {stt[stt != 0][:4]} (rank_cap / footprint)")
        self.stats["device_steps"] += n
        self.stats["launches"] += 1
        self.t_num[parents, ks] = r * (costs + 1.0)
        if st.any():
            c = kids[st]
            pp = self.n_devpath[parents[st]].copy()
            depth_of = (pp >= 0).sum(axis=1)
            pp[np.arange(len(c)), depth_of] = new_ids[st]
            self.n_devpath[c] = pp
            self.n_dev[c] = new_ids[st]

    def _vexpand(self, nd, idx, K, mask, policy, value, noise, rng):
        """mcts.py:204-233 (BatchedMCTS._expand_leaf) into the tables."""
        A = self.num_actions
        vi = idx[:K]
        if mask is not None or policy is not None and len(np.atleast_1d(policy)) == A:
            if mask is None:
                mask = np.zeros(A, dtype=bool)
                mask[vi] = True
            full = (np.ones(A) / A if policy is None else np.asarray(policy, dtype=np.float64)) * mask
            if noise:
                full = (1 - self.eps) * full + self.eps * rng.dirichlet([self.alpha] * A)
                full = full / np.sum(full)
            total = np.sum(full)
            ps = full[vi]
        else:
            ps = np.full(K, 1.0 / A) if policy is None else np.asarray(policy, dtype=np.float64)[:K]  # sparse policy: on valid_idx
            if noise:
                # Dirichlet(alpha) over all A actions, looked at on the K valid ones: independent Gamma(alpha) draws for those,
                # one Gamma((A - K) alpha) draw for the total of the rest (aggregation property).  The reference normalises the
                # noisy vector over ALL actions: the mass that lands on invalid actions stays there (mcts.py:160-164,222-225)
                g = rng.gamma(self.alpha, size=K)
                rest = rng.gamma(self.alpha * (A - K)) if A > K else 0.0
                ps = ((1 - self.eps) * ps + self.eps * g / (g.sum() + rest)) / ((1 - self.eps) * float(np.sum(ps)) + self.eps)
                total = 1.0
            else:
                total = float(np.sum(ps))
        ps = ps / total if total > 0 else np.full(K, 1.0 / K)
        self.t_idx[nd, :K] = vi
        self.t_Ps[nd, :K] = ps
        self.n_K[nd] = K
        self.n_Ns[nd] = 0
        self.n_expanded[nd] = True
        self.n_value[nd] = value

    def _policy_sparse(self, root, prev, budget, temperature, deploy_time, rng):
        """mcts.py:98-143 on the root's valid actions; returns (policy, second) like the reference for small action sets and
        ({action index: probability}, valid indices) for large ones."""
        K = int(self.n_K[root])
        if not self.n_expanded[root] or K == 0:
            return None
        idx = self.t_idx[root, :K]
        nsa = self.t_Nsa[root, :K].copy()
        visits = nsa.copy()
        ps, ns = self.t_Ps[root, :K], self.n_Ns[root]
        if not deploy_time:
            best = int(rng.choice((visits == np.max(visits)).nonzero()[0])) if visits.max() > 0 or self.num_actions == K else None
            if best is None:  # every valid action unvisited: the reference's argmax runs over ALL actions (zeros) -- no pruning possible
                best = -1
            nfp = np.ceil(np.sqrt(self.fpf * ps * ns))
            nfp[visits == 0] = 0
            uct = self._uct_rows(np.array([root]), force_playouts=False)[0, :K]
            max_puct = uct[best] if best >= 0 else -np.inf
            q = self.t_Qsa[root, :K]
            qn = self._normalize_q(q, K < self.num_actions)
            for k in range(K):
                if k == best or nfp[k] <= 0:
                    continue
                for _ in range(int(nfp[k])):
                    visits[k] -= 1
                    prior = self.puct_init + np.log((ns + self.puct_base + 1) / self.puct_base)
                    prior *= ps[k] * (np.sqrt(ns + 1) / (1 + visits[k]))
                    if qn[k] + prior >= max_puct:
                        visits[k] += 1
                        break
            visits[visits == 1] = 0
        if np.sum(visits) == 0:
            return None
        dense = self.num_actions <= self.DENSE_ACTIONS
        if temperature == 0:
            b = int(rng.choice(np.array(np.argwhere(visits == np.max(visits))).flatten()))
            if dense:
                policy = [0] * self.num_actions
                policy[int(idx[b])] = 1
                full = np.zeros(self.num_actions)
                full[idx] = visits
                return policy, full
            return {int(idx[b]): 1.0}, idx
        vt = visits ** (1.0 / temperature)
        p = vt / np.sum(vt)
        if dense:
            full = np.zeros(self.num_actions)
            full[idx] = p
            return full.tolist(), self.next_actions_mask(prev, budget)
        return {int(i): float(x) for i, x in zip(idx, p) if x > 0}, idx
