"""
DeviceMCTS: the tree search of mcts.py / vector_mcts.py with selection, expansion and backup ON THE GPU (csrc/k_mcts.h).

VectorMCTS spends 95 % of a BASELINE configs[4] search (1024 roots x 256 simulations, 200x200) in NumPy: 5 s around 0.1 s
of device work.  Here the node tables live in HBM (torch tensors handed to the C-ABI as ipp_mcts_tables), one wavefront
owns one root, and the host only sequences the launches of a wave of simulations:

    ipp_mcts_select            W descents per root (PUCT, forced playouts, virtual visits, transposition lookup)
    (one small read-back, under the step launch: the number of covariance steps requested and whether any leaf is pending)
    ipp_mcts_steps             ipp_tree_step on the wave's request list (ONE launch: the requests are independent) + edge numerators
    ipp_mcts_expand            valid-action sets, priors (uniform or the network's), Dirichlet noise at the root
    ipp_mcts_backup            values back along the recorded descents

Same search as VectorMCTS(sims_in_flight=W) -- same formulas in the same floating-point order, same node identity (the
commutative 64-bit key of the action multiset), same order of simulations and backups: with tie_break="first" both build
the same trees (tests/test_hip_mcts.py).  Differences: the Dirichlet noise and tie_break="random" draw from counter-based
streams on the device (statistically equivalent, not NumPy's streams), and the network is asked with TENSORS:

    infer(batch) -> (prior or None, value)
        batch: dict of device tensors for the n pending leaves of this wave -- "root" (env slot), "node", "depth",
        "previous_action" [n, 3], "budget" [n], "valid_idx" [n, kmax] (ascending action indices, -1 padded), "K" [n]
        prior: [n, kmax] float64 probabilities on the valid sets (need not be normalised) or None = uniform
        value: [n] float64 tensor or a float
    infer=None: uniform priors, value `leaf_value` (the stub of the benchmarks).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from ... import _ffi
from .vector_mcts import VectorMCTS


class DeviceMCTS(VectorMCTS):
    def __init__(self, engine, hyper_params: Dict, meta_data: Dict, infer: Optional[Callable] = None, sims_in_flight: int = 4,
                 tie_break: str = "first", seed: int = 0, leaf_value: float = 0.0, nodes_per_root: Optional[int] = None,
                 dev_per_root: Optional[int] = None, queue_ahead: bool = True, groups: int = 2):
        """groups > 1: the roots of a search are split into that many contiguous groups, each with its own node tables, whose waves of
        simulations alternate on their own streams -- one group's selection (one wavefront per SIMD: latency-bound) runs beside the
        other's tree steps, and the host's one read-back per wave of simulations waits for one group while the other's launches are
        queued.  Same trees and policies as groups = 1 (the counter-based draws are keyed on a root's number in the whole search).
        Used when the read-out runs on the device (temperature > 0, rngs=None, a large action set) and the engine's max_batch holds
        all roots x sims_in_flight items; otherwise the search runs in one piece."""
        super().__init__(engine, hyper_params, meta_data, infer, None, sims_in_flight, tie_break, seed)
        self.groups = max(1, int(groups))
        self._ctor = dict(hyper_params=hyper_params, meta_data=meta_data, infer=infer, sims_in_flight=sims_in_flight, tie_break=tie_break,
                          seed=seed, leaf_value=leaf_value, nodes_per_root=nodes_per_root, dev_per_root=dev_per_root, queue_ahead=queue_ahead)
        self._root_base = self._dev_base = self._scratch_base = 0  # (a group's offsets in the whole search: ipp_mcts_tables)
        self._total_roots = None
        self._subs = None
        self._group_streams = None
        self.leaf_value = float(leaf_value)
        self.queue_ahead = bool(queue_ahead)  # (False: every step launch waits for the wave's request count -- tests, A/B)
        self.seed = int(seed)
        S, W = self.num_simulations, self.sims_in_flight
        # a descent passes through at most horizon + 1 edges and only its first traversal of an edge creates a node; more
        # than one new node per simulation needs transpositions into expanded nodes, hence the slack (overflow is an error)
        self.nodes_per_root = int(nodes_per_root or (2 * S + 2 * W + 8))
        self.dev_per_root = int(dev_per_root or max(1, min(self.nodes_per_root, engine._c.node_capacity // max(1, engine.capacity))))
        self._tab = None
        self._tab_roots = 0

    # ------------------------------------------------------------------ buffers
    def _geometry(self, torch, dev):
        if getattr(self, "_geo", None) is not None:
            return self._geo
        W_, H_ = self._W, self._H
        ca = self._cell_action
        # the action index is affine in the cell: order the offsets so that it ascends (valid sets come out sorted)
        a = int(ca[1, 0] - ca[0, 0]) if W_ > 1 else 0
        b = int(ca[0, 1] - ca[0, 0]) if H_ > 1 else 0
        gx, gy = np.meshgrid(np.arange(W_), np.arange(H_), indexing="ij")
        if not np.array_equal(ca, ca[0, 0] + a * gx + b * gy):
            raise ValueError("action enumeration is not affine in the cell index")
        order = np.argsort(a * self._off_x + b * self._off_y, kind="stable")
        kmax = int(min(self.Kmax, self.num_actions))
        x = 1.0 / self.num_actions
        uniform = np.array([0.0] + [x / float(np.sum(np.full(k, x))) for k in range(1, kmax + 1)])
        t = lambda arr, dt: torch.as_tensor(np.ascontiguousarray(arr), dtype=dt, device=dev)  # noqa: E731
        self._geo = dict(
            kmax=kmax,
            actions=t(self.actions_np, torch.float64), cell_action=t(ca, torch.int32),
            off_x=t(self._off_x[order], torch.int32), off_y=t(self._off_y[order], torch.int32),
            zkey=t(self._z.view(np.int64), torch.int64), uniform=t(uniform, torch.float64))
        return self._geo

    def _alloc(self, R: int, D: int):
        import torch

        eng = self.engine
        dev = eng.device
        geo = self._geometry(torch, dev)
        K, npr, W = geo["kmax"], self.nodes_per_root, self.sims_in_flight
        if self._dev_base + R * self.dev_per_root > eng._c.node_capacity:
            raise ValueError(f"{R} roots x {self.dev_per_root} device nodes (from node {self._dev_base}) exceed the engine's node_capacity {eng._c.node_capacity}")
        cap = R * npr
        tsz = 1
        while tsz < 2 * npr:
            tsz *= 2
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
        b = dict(
            t_idx=e((cap, K), torch.int32), t_ps=e((cap, K), torch.float64), t_nsa=e((cap, K), torch.float64),
            t_qsa=e((cap, K), torch.float64), t_num=e((cap, K), torch.float64), t_child=e((cap, K), torch.int32),
            n_k=e((cap,), torch.int32), n_ns=e((cap,), torch.float64), n_flags=e((cap,), torch.uint8), n_hash=e((cap,), torch.int64),
            n_value=e((cap,), torch.float64), n_devpath=e((cap, 6), torch.int32), root_count=e((R,), torch.int32),
            dev_count=e((R,), torch.int32), h_keys=e((R, tsz), torch.int64), h_vals=e((R, tsz), torch.int32),
            p_node=e((W, R, D), torch.int32), p_k=e((W, R, D), torch.int32), p_cost=e((W, R, D), torch.float64),
            p_len=e((W, R), torch.int32), leaf=e((W, R), torch.int32),
            pend_node=e((R, W), torch.int32), pend_depth=e((R, W), torch.int32), pend_sim=e((R, W), torch.int32),
            pend_prev=e((R, W, 3), torch.float64), pend_budget=e((R, W), torch.float64),
            # pend_count [R] and rq_count [1] share one buffer: cleared and read back together
            counts=e((R + 1,), torch.int32),
            # (one request list per wave of simulations; a descent asks for at most D steps, typically one)
            rq_root=e((D * R * W,), torch.int32), rq_parent=e((D * R * W,), torch.int32), rq_k=e((D * R * W,), torch.int32),
            rq_child=e((D * R * W,), torch.int32), rq_newdev=e((D * R * W,), torch.int32), rq_cost=e((D * R * W,), torch.float64),
            rq_prev=e((D * R * W, 3), torch.float64), rq_action=e((D * R * W, 3), torch.float64),
            ts_paths=e((D * R * W, 6), torch.int32), ts_reward=e((D * R * W,), torch.float32), ts_status=e((D * R * W,), torch.int32),
            err=e((4,), torch.int32),
        )
        uav = self.uav
        tab = _ffi.IppMctsTables(
            roots=R, kmax=K, nodes_per_root=npr, dev_per_root=self.dev_per_root, table_size=tsz, max_depth=D, wave=W,
            horizon=self.horizon, grid_w=self._W, grid_h=self._H, n_levels=self._n_lv, n_off=len(self._off_x),
            num_actions=self.num_actions, use_flight_time=1 if uav is not None else 0, tie_break=0 if self.tie_break == "first" else 1,
            device=dev.index or 0, res=self._res, max_dist=self.max_dist, gamma=self.gamma, puct_init=self.puct_init,
            puct_base=self.puct_base, fpf=self.fpf, vmax=float(uav["max_v"]) if uav else 1.0, amax=float(uav["max_a"]) if uav else 1.0,
            root_base=self._root_base, dev_base=self._dev_base, scratch_base=self._scratch_base)
        for name in ("actions", "cell_action", "off_x", "off_y", "zkey"):
            setattr(tab, name, geo[name].data_ptr())
        tab.uniform_ps = geo["uniform"].data_ptr()
        for name, buf in b.items():
            if name != "counts":
                setattr(tab, name, buf.data_ptr())
        # the Ns-only factors of the PUCT prior, tabulated with NumPy (the arithmetic VectorMCTS._uct_rows does per node): a node is
        # visited at most once per simulation, virtual visits included
        ns = np.arange(self.num_simulations + W + 8, dtype=np.float64)
        self._ns_tables = (torch.as_tensor(self.puct_init + np.log((ns + self.puct_base + 1) / self.puct_base), device=dev),
                           torch.as_tensor(np.sqrt(ns + 1), device=dev))
        tab.puct_c, tab.sqrt_ns1, tab.ns_table_n = self._ns_tables[0].data_ptr(), self._ns_tables[1].data_ptr(), len(ns)
        tab.pend_count = b["counts"].data_ptr()
        tab.rq_count = b["counts"].data_ptr() + 4 * R
        self._tab, self._buf, self._tab_roots, self._tab_depth = tab, b, R, D
        # read-out (ipp_mcts_policy): results on the device and their pinned host copies
        self._out = dict(policy=e((R, K), torch.float64), valid_idx=e((R, K), torch.int32), ok=e((R,), torch.int32), u=e((R,), torch.float64))
        pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)  # noqa: E731
        self._counts_host, self._counts_ev = pin((R + 1,), torch.int32), torch.cuda.Event()
        self._out_host = dict(policy=pin((R, K), torch.float64), valid_idx=pin((R, K), torch.int32), ok=pin((R,), torch.int32),
                              K=pin((R,), torch.int32), nodes=pin((), torch.int64))
        return tab, b

    # ------------------------------------------------------------------ search
    def get_policy(self, roots: Sequence[int], previous_actions, budgets, depth: int = 0, temperature: float = 1.0,
                   deploy_time: bool = False, rngs=None, as_arrays: bool = False):
        """Like VectorMCTS.get_policy: one search of num_mcts_simulations per root, policies from the roots' visit counts.

        The read-out (mcts.py:83-143) runs on the device too (ipp_mcts_policy) when temperature > 0, the action set is a large one and
        the roots share one generator (rngs=None); the reference's per-root (dict, valid indices) pairs are then built from two
        arrays.  as_arrays=True skips that: returns {"policy": [R, kmax] float64, "valid_idx": [R, kmax] int32 (-1 padded),
        "K": [R], "ok": [R] (0 where the reference returns None)} as DEVICE tensors (valid until the next search)."""
        R = len(roots)
        on_device = (rngs is None and temperature > 0 and self.num_actions > self.DENSE_ACTIONS
                     and os.environ.get("IPP_MCTS_HOST_READOUT", "0") != "1")
        G = min(self.groups, R // 2)
        if G > 1 and on_device and self.queue_ahead and bool(int(self.engine.info.patch_layout)) and \
                self.engine.max_batch >= R * self.sims_in_flight:
            return self._get_policy_groups(G, roots, previous_actions, budgets, depth, temperature, deploy_time, as_arrays)
        self._subs_used = None
        state = {}
        for _ in self._search(roots, previous_actions, budgets, depth, state):
            pass  # (one piece: wait where the generator says the host has to)
        return self._policies(state["b"], R, self.nodes_per_root, state["prev0"], state["budget0"], temperature, deploy_time, rngs, roots, as_arrays)

    def _search(self, roots, previous_actions, budgets, depth, state):
        """The waves of simulations of one search on the CURRENT stream, as a generator: it yields right before the host has to wait
        for the wave's counts, so that a caller with several groups of roots can issue the other groups' launches first."""
        import torch

        eng, lib = self.engine, self.engine._lib
        dev = eng.device
        R = len(roots)
        D = max(1, self.horizon + 1 - depth)
        W = self.sims_in_flight
        if self._tab is None or self._tab_roots != R or self._tab_depth != D:
            self._alloc(R, D)
        tab, b = self._tab, self._buf
        prev0 = torch.as_tensor(np.asarray(previous_actions, dtype=np.float64).reshape(R, 3), device=dev)
        budget0 = torch.as_tensor(np.asarray(budgets, dtype=np.float64).reshape(R), device=dev)
        root_env = torch.as_tensor(np.asarray(roots, dtype=np.int32), device=dev)
        npr = self.nodes_per_root
        root_nodes = torch.arange(R, device=dev, dtype=torch.int64) * npr
        # ---- start of a search (include/ipp_engine.h: contract of ipp_mcts_tables)
        b["n_flags"].zero_()
        b["n_flags"][root_nodes] = 2
        b["n_value"].zero_()
        b["n_devpath"].fill_(-1)
        b["n_hash"][root_nodes] = (torch.arange(R, device=dev, dtype=torch.int64) + 1 + self._root_base) * (-7046029254386353131)  # 0x9E3779B97F4A7C15
        b["root_count"].fill_(1)
        b["dev_count"].zero_()
        b["h_keys"].zero_()
        b["err"].zero_()
        b["counts"].zero_()  # (pending leaves / requests of a wave of simulations: ipp_mcts_backup clears them again)
        flags = (_ffi.IPP_ADAPTIVE if self.adaptive else 0) | (_ffi.IPP_USE_FLIGHT_TIME if self.uav is not None else 0)
        stream = eng.stream
        tp = C.byref(tab)
        sim = 0
        # A wave of simulations: select, the covariance steps it asked for (ONE list: they do not depend on each other), expand,
        # backup.  With tree nodes stored as patches (ipp_info.patch_layout: the tree-step kernel reads its item count on the device)
        # the step launch is queued right behind the selection, sized for roots x wave items (n = -1); the host reads the counts
        # (pending leaves, requests) into pinned memory meanwhile and only launches what the list holds beyond that size (descents
        # that pass several new edges through transpositions: a few dozen of 8192 at configs[4]).  Band-tile engines (and
        # queue_ahead=False) launch the exact count after the read-back.
        ahead = self.queue_ahead and bool(int(eng.info.patch_layout)) and eng.max_batch >= R * W
        counts_h, ev = self._counts_host, self._counts_ev
        while sim < self.num_simulations:
            w = min(W, self.num_simulations - sim)
            _ffi.check(lib.ipp_mcts_select(tp, root_env.data_ptr(), prev0.data_ptr(), budget0.data_ptr(), int(depth), int(sim), int(w),
                                           C.c_uint64(self.seed & (2 ** 64 - 1)), stream))
            counts_h.copy_(b["counts"], non_blocking=True)
            ev.record()
            if ahead:
                _ffi.check(lib.ipp_mcts_steps(eng._h, tp, 0, -1, flags, stream))
            yield sim
            ev.synchronize()  # the one synchronisation of the wave
            n, n_pending = int(counts_h[R]), int(counts_h[:R].sum())
            first = R * W if ahead else 0
            while first < n:
                chunk = min(n - first, eng.max_batch)
                _ffi.check(lib.ipp_mcts_steps(eng._h, tp, first, chunk, flags, stream))
                first += chunk
                self.stats["launches"] += 1
            self.stats["device_steps"] += n
            self.stats["launches"] += 1 if ahead else 0
            if n_pending:
                self._expand(lib, tp, b, R, W, root_env, stream)
                self.stats["inferences"] += n_pending
            _ffi.check(lib.ipp_mcts_backup(tp, int(w), stream))
            sim += w
        err = b["err"].cpu().numpy()
        if err[0] or err[1] or err[3]:
            raise RuntimeError(f"device tree search ran out of room (nodes, device nodes, kmax) = {err[0], err[1], err[3]}: "
                               f"raise nodes_per_root / the engine's node_capacity")
        if err[2]:
            raise RuntimeError(f"ipp_tree_step reported status {int(err[2])} (rank_cap / footprint)")
        state.update(b=b, prev0=prev0, budget0=budget0)

    # ------------------------------------------------------------------ groups of roots on their own streams
    def _group_setup(self, G, R):
        """Sub-searches (one table set each) for the contiguous groups of R roots and one stream per group, on different hardware
        queues where the runtime has them (ipp_probe_stream_pair, as VecIPPEnv picks the streams of its parts)."""
        import torch

        sizes = [R // G + (1 if g < R % G else 0) for g in range(G)]
        if self._subs is None or [s._group_size for s in self._subs] != sizes:
            self._subs = []
            base = 0
            for g, n in enumerate(sizes):
                sub = DeviceMCTS(self.engine, groups=1, **self._ctor)
                sub.nodes_per_root, sub.dev_per_root = self.nodes_per_root, self.dev_per_root
                sub._root_base, sub._dev_base, sub._scratch_base = base, base * self.dev_per_root, base * self.sims_in_flight
                sub._total_roots, sub._group_size = R, n
                sub.stats = self.stats  # (one set of counters for the whole search)
                self._subs.append(sub)
                base += n
        if self._group_streams is None or len(self._group_streams) != G:
            eng, main = self.engine, torch.cuda.current_stream(self.engine.device)
            got = []
            try:
                thr = 0.75 * min(eng.probe_stream_pair(main, main, 12) for _ in range(2))
                for _ in range(12):
                    st = torch.cuda.Stream(device=eng.device)
                    with torch.cuda.stream(st):
                        torch.zeros(8, device=eng.device).add_(1)
                    if eng.probe_stream_pair(st, main, 12) > thr:
                        continue  # (shares the caller's queue)
                    if all(eng.probe_stream_pair(st, o, 12) <= thr for o in got):
                        got.append(st)
                    if len(got) == G:
                        break
            except Exception:
                got = []
            while len(got) < G:  # (fewer hardware queues than groups, or no probe: streams as they come)
                got.append(torch.cuda.Stream(device=eng.device))
            self._group_streams = got
        return sizes

    def _get_policy_groups(self, G, roots, previous_actions, budgets, depth, temperature, deploy_time, as_arrays):
        import torch

        R = len(roots)
        sizes = self._group_setup(G, R)
        prev = np.asarray(previous_actions, dtype=np.float64).reshape(R, 3)
        bud = np.asarray(budgets, dtype=np.float64).reshape(R)
        main = torch.cuda.current_stream(self.engine.device)
        gens, states, lo = [], [], 0
        for sub, n, st in zip(self._subs, sizes, self._group_streams):
            st.wait_stream(main)  # (the roots' states were written on the caller's stream)
            state = {}
            gens.append(sub._search(list(roots[lo:lo + n]), prev[lo:lo + n], bud[lo:lo + n], depth, state))
            states.append(state)
            lo += n
        alive = list(range(G))
        while alive:  # a turn = everything a group can issue before its next wait: the other groups' launches are queued by then
            for g in list(alive):
                with torch.cuda.stream(self._group_streams[g]):
                    try:
                        next(gens[g])
                    except StopIteration:
                        alive.remove(g)
        outs = []
        for sub, n, st, state in zip(self._subs, sizes, self._group_streams, states):
            with torch.cuda.stream(st):
                outs.append(sub._policies(state["b"], n, self.nodes_per_root, state["prev0"], state["budget0"], temperature, deploy_time, None,
                                          None, as_arrays))
            main.wait_stream(st)
        self.stats["nodes"] = sum(sub._nodes_last for sub in self._subs)
        self._subs_used = self._subs
        self._host_rows = {}
        self.root_ids = np.arange(R)
        if as_arrays:
            return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
        return [p for o in outs for p in o]

    def _expand(self, lib, tp, b, R, W, root_env, stream):
        import torch

        seed = C.c_uint64((self.seed * 0x9E3779B97F4A7C15 + 12345) & (2 ** 64 - 1))
        if self.infer is None:
            _ffi.check(lib.ipp_mcts_expand(tp, None, None, self.leaf_value, 0, self.alpha, self.eps, seed, stream))
            return
        # network: valid sets first, then the replies scattered into [R W] slot order
        _ffi.check(lib.ipp_mcts_expand(tp, None, None, 0.0, 1, self.alpha, self.eps, seed, stream))
        cnt = b["counts"][:R].to(torch.int64)
        slot = torch.arange(W, device=cnt.device)[None, :] < cnt[:, None]  # [R, W] pending slots
        g = torch.nonzero(slot.reshape(-1)).reshape(-1)
        nodes = b["pend_node"].reshape(-1)[g].to(torch.int64)
        batch = dict(root=root_env[(g // W)], node=nodes, depth=b["pend_depth"].reshape(-1)[g],
                     previous_action=b["pend_prev"].reshape(-1, 3)[g], budget=b["pend_budget"].reshape(-1)[g],
                     valid_idx=b["t_idx"][nodes], K=b["n_k"][nodes])
        prior, value = self.infer(batch)
        K = b["t_idx"].shape[1]
        pr_all = None
        if prior is not None:
            pr_all = torch.zeros((R * W, K), dtype=torch.float64, device=cnt.device)
            pr_all[g] = torch.as_tensor(prior, dtype=torch.float64, device=cnt.device).reshape(len(g), K)
        if torch.is_tensor(value) or isinstance(value, np.ndarray):
            v_all = torch.zeros((R * W,), dtype=torch.float64, device=cnt.device)
            v_all[g] = torch.as_tensor(value, dtype=torch.float64, device=cnt.device).reshape(len(g))
            v_ptr, v_const = v_all.data_ptr(), 0.0
        else:
            v_all, v_ptr, v_const = None, None, float(value)
        _ffi.check(lib.ipp_mcts_expand(tp, pr_all.data_ptr() if pr_all is not None else None, v_ptr, v_const, 0, self.alpha, self.eps,
                                       seed, stream))
        self._keep_infer = (pr_all, v_all)

    def _policies(self, b, R, npr, prev0, budget0, temperature, deploy_time, rngs, roots, as_arrays=False):
        """get_policy (mcts.py:83-143) from the root rows: on the device (ipp_mcts_policy), or through VectorMCTS._policy_sparse /
        _policies_rows on host copies of the R root rows (small action sets, temperature 0, per-root generators)."""
        self._host_rows = {}  # (the root rows come to the host when somebody asks for them: t_idx, t_Nsa, ... below)
        self._rows_src = (b, R, npr)
        self.root_ids = np.arange(R)
        self._shared_rng = None
        on_device = (rngs is None and temperature > 0 and self.num_actions > self.DENSE_ACTIONS
                     and os.environ.get("IPP_MCTS_HOST_READOUT", "0") != "1")
        if on_device:
            return self._policies_device(b, R, npr, temperature, deploy_time, as_arrays)
        if as_arrays:
            raise ValueError("as_arrays needs the read-out on the device: temperature > 0, rngs=None and more than DENSE_ACTIONS actions")
        self.stats["nodes"] = int((b["n_flags"] & 1).sum().item())
        prev0, budget0 = prev0.cpu().numpy(), budget0.cpu().numpy()
        if rngs is None:  # one generator for the read-out draws of all roots (1024 seeded RandomStates cost 50 ms)
            shared = np.random.RandomState(self.seed & 0x7fffffff)
            rngs = [shared] * R
            self._shared_rng = shared
        else:
            rngs = list(rngs)
        if self.num_actions <= self.DENSE_ACTIONS or temperature == 0:
            return [self._policy_sparse(j, prev0[j], float(budget0[j]), temperature, deploy_time, rngs[j]) for j in range(R)]
        return self._policies_rows(R, temperature, deploy_time, rngs)

    def _policies_device(self, b, R, npr, temperature, deploy_time, as_arrays):
        """ipp_mcts_policy on the root rows; the kept action among equally visited ones is picked by the same draw of the shared
        generator as in _policies_rows (one uniform per root)."""
        import torch

        eng, o = self.engine, self._out
        u_ptr = None
        if not deploy_time:
            shared = np.random.RandomState(self.seed & 0x7fffffff)
            self._shared_rng = shared
            # (a group of a split search takes ITS roots' draws of the whole search's sequence)
            u_all = shared.random_sample(self._total_roots or R)
            o["u"].copy_(torch.from_numpy(u_all[self._root_base:self._root_base + R]))
            u_ptr = o["u"].data_ptr()
        _ffi.check(eng._lib.ipp_mcts_policy(C.byref(self._tab), u_ptr, float(temperature), int(bool(deploy_time)), o["policy"].data_ptr(),
                                            o["valid_idx"].data_ptr(), o["ok"].data_ptr(), eng.stream))
        nodes = (b["n_flags"] & 1).sum()
        root_nodes = torch.arange(R, device=o["ok"].device, dtype=torch.int64) * npr
        K_dev = b["n_k"][root_nodes]
        if as_arrays:
            self._nodes_last = self.stats["nodes"] = int(nodes.item())
            return dict(policy=o["policy"], valid_idx=o["valid_idx"], K=K_dev, ok=o["ok"])
        h = self._out_host
        h["policy"].copy_(o["policy"], non_blocking=True)
        h["valid_idx"].copy_(o["valid_idx"], non_blocking=True)
        h["ok"].copy_(o["ok"], non_blocking=True)
        h["K"].copy_(K_dev, non_blocking=True)
        h["nodes"].copy_(nodes, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self._nodes_last = self.stats["nodes"] = int(h["nodes"].item())
        p, ok, K = h["policy"].numpy(), h["ok"].numpy(), h["K"].numpy()
        t_idx = h["valid_idx"].numpy().astype(np.int64)
        self._host_rows["t_idx"] = t_idx
        self._host_rows["n_K"] = K.astype(np.int64)
        out = []
        for j in range(R):
            if not ok[j]:
                out.append(None)
                continue
            idx = t_idx[j, :K[j]]
            pj = p[j, :K[j]]
            nz = pj > 0
            out.append((dict(zip(idx[nz].tolist(), pj[nz].tolist())), idx))
        return out

    def _policies_rows(self, R, temperature, deploy_time, rngs):
        """VectorMCTS._policy_sparse for all roots at once (large action sets, temperature > 0): the same arithmetic on
        [R, kmax] arrays -- forced-playout pruning (mcts.py:109-131) as a masked loop over the playouts to take back -- and the
        same draws from the per-root generators; only the Python objects are built per root."""
        valid = self.t_idx >= 0
        K = self.n_K
        visits = np.where(valid, self.t_Nsa, 0.0)
        ok_root = self.n_expanded & (K > 0)
        if not deploy_time:
            vmax = visits.max(axis=1)
            best = np.full(R, -1, dtype=np.int64)
            if self._shared_rng is not None:
                # all roots draw from one generator: one uniform per root picks among its most-visited actions (the per-root
                # rng.choice loop was ~10 of the 45 ms of a 1024-root search)
                ties = (visits == vmax[:, None]) & (np.arange(visits.shape[1])[None, :] < K[:, None])
                n_ties = ties.sum(axis=1)
                pick = np.minimum((self._shared_rng.random_sample(R) * n_ties).astype(np.int64), np.maximum(n_ties - 1, 0))
                order = np.cumsum(ties, axis=1) - 1
                chosen = (ties & (order == pick[:, None])).argmax(axis=1)
                take = ok_root & (n_ties > 0) & ((vmax > 0) | (self.num_actions == K))
                best[take] = chosen[take]
            else:
                for j in np.nonzero(ok_root)[0]:
                    if vmax[j] > 0 or self.num_actions == K[j]:
                        best[j] = int(rngs[j].choice((visits[j, :K[j]] == vmax[j]).nonzero()[0]))
            ps, ns = self.t_Ps, self.n_Ns[:, None]
            with np.errstate(invalid="ignore"):
                nfp = np.ceil(np.sqrt(self.fpf * ps * ns))
            nfp[(visits == 0) | ~valid] = 0
            uct = self._uct_rows(np.arange(R), force_playouts=False)
            max_puct = np.where(best >= 0, uct[np.arange(R), np.maximum(best, 0)], -np.inf)[:, None]
            # Q normalised like _normalize_q (the zeros of the invalid actions take part when the set is a strict subset)
            q = np.where(valid, self.t_Qsa, 0.0)
            has_outside = (K < self.num_actions)[:, None]
            lo = np.where(valid, q, np.inf).min(axis=1, keepdims=True)
            hi = np.where(valid, q, -np.inf).max(axis=1, keepdims=True)
            lo = np.where(has_outside, np.minimum(lo, 0.0), lo)
            hi = np.where(has_outside, np.maximum(hi, 0.0), hi)
            allzero = np.all(q == 0, axis=1, keepdims=True)
            with np.errstate(divide="ignore", invalid="ignore"):
                qn = np.where(allzero, q, np.where(lo == hi, q / hi, (q - lo) / (hi - lo)))
            pc = self.puct_init + np.log((ns + self.puct_base + 1) / self.puct_base)
            live = valid & (nfp > 0) & (np.arange(visits.shape[1])[None, :] != best[:, None])
            left = np.where(live, nfp, 0.0)
            while live.any():  # one forced playout taken back per pass; an action leaves when its PUCT reaches the best's
                visits = np.where(live, visits - 1, visits)
                with np.errstate(divide="ignore", invalid="ignore"):
                    prior = pc * (ps * (np.sqrt(ns + 1) / (1 + visits)))
                back = live & (qn + prior >= max_puct)
                visits = np.where(back, visits + 1, visits)
                left = left - 1
                live = live & ~back & (left > 0)
            visits[visits == 1] = 0
        out = []
        tot = visits.sum(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            vt = visits ** (1.0 / temperature)
            p = vt / vt.sum(axis=1, keepdims=True)
        for j in range(R):
            if not ok_root[j] or tot[j] == 0:
                out.append(None)
                continue
            idx = self.t_idx[j, :K[j]]
            pj = p[j, :K[j]]
            nz = pj > 0
            out.append((dict(zip(idx[nz].tolist(), pj[nz].tolist())), idx))
        return out

    # The root rows on the host (what VectorMCTS keeps as NumPy tables): copied from the device when first asked for after a search.
    _ROOT_ROWS = {"t_idx": ("t_idx", lambda a: a.astype(np.int64)), "t_Ps": ("t_ps", None), "t_Nsa": ("t_nsa", None), "t_Qsa": ("t_qsa", None),
                  "n_K": ("n_k", lambda a: a.astype(np.int64)), "n_Ns": ("n_ns", None), "n_expanded": ("n_flags", lambda a: (a & 1).astype(bool))}

    def _root_rows(name):  # noqa: N805 (builds the properties below)
        def get(self):
            rows = self.__dict__.setdefault("_host_rows", {})
            subs = self.__dict__.get("_subs_used")
            if name not in rows and subs:
                rows[name] = np.concatenate([getattr(sub, name) for sub in subs])
            if name not in rows:
                src = self.__dict__.get("_rows_src")
                if src is None:
                    raise AttributeError(f"{name}: no search yet")
                import torch

                b, R, npr = src
                key, conv = DeviceMCTS._ROOT_ROWS[name]
                a = b[key][torch.arange(R, device=b[key].device, dtype=torch.int64) * npr].cpu().numpy()
                rows[name] = conv(a) if conv else a
            return rows[name]

        def put(self, value):
            self.__dict__.setdefault("_host_rows", {})[name] = value

        return property(get, put)

    t_idx, t_Ps, t_Nsa, t_Qsa = _root_rows("t_idx"), _root_rows("t_Ps"), _root_rows("t_Nsa"), _root_rows("t_Qsa")
    n_K, n_Ns, n_expanded = _root_rows("n_K"), _root_rows("n_Ns"), _root_rows("n_expanded")
    del _root_rows

    def root_statistics(self):
        """(valid action indices, visit counts, Q) of every root after get_policy: arrays [R, kmax], padding idx -1."""
        return self.t_idx, self.t_Nsa, self.t_Qsa
