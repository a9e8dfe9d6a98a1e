"""
GPU tests at the sizes BASELINE.json's configs name, beyond configs[1] (tests/test_hip_edge_cases.py):

  configs[2]  32768 envs, 100x100 grid, 16-step episodes            (arena ~200 GB, factor state, window from the prior)
  configs[3]  262144 envs over 8 GPUs = 32768 envs of 50x50 per GPU  (arena ~125 GB), 40-step episodes
  configs[4]  1024 root states x 256 simulations on a 200x200 grid   (ipp_tree_step on path-local columns)

The fp64 oracle cannot run these batches, so the full-size runs are checked three ways: (1) invariants on every env
(status 0, reward >= 0 and finite, trace strictly decreasing, ranks as expected), (2) the first envs of the batch
against the EXACT factor mode / the factor-form oracle fed the same inputs (envs are independent: env e of the big
batch must equal env e of a small one), (3) a second identical run must reproduce every reward bit for bit.
The small-batch, cell-for-cell oracle comparisons at these grid sizes are in test_hip_big_grids.py (100x100, 200x200)
and below (tree steps at 200x200, depth 1-6).
"""
import copy
import gc

import numpy as np
import pytest

from oracle import ipp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
UAV = {"max_v": 2.0, "max_a": 2.0}
ALTS = [float(a) for a in range(5, 15)]
D = 6


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def pad(path):
    return list(path) + [-1] * (D - len(path))


def fresh_gpu():
    import torch

    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("grid,B,T,name", [(100, 32768, 16, "configs[2]"), (50, 32768, 40, "configs[3] per-GPU share")])
def test_fullsize_batch_invariants_subset_parity_and_reproducibility(grid, B, T, name):
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.vec_env import cell_centre_actions

    fresh_gpu()
    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    ocfg = orc.OracleConfig(x_dim=grid, y_dim=grid)
    N, S = cfg.n_cells, 192  # S: envs also run through the exact factor mode
    full = IPPEngine(cfg, capacity=B, state="factor", rank_cap=9 * T, window_rows=-1, fixed_prior=True)
    exact = IPPEngine(cfg, capacity=S, state="factor", rank_cap=9 * T, window_rows=0)
    # the bench's path: window from the fixed prior, factor columns as compact patches, k_step_patch -- and for launches of this size
    # its six-waves-per-SIMD instantiation (csrc/ipp_engine.hip patch_layout() / launch_chunk()); a regression in the selection rules
    # would put these configs back on the band-tile kernels with every numeric check below still green
    assert full.info.window_rows == 10
    assert full.info.patch_layout == 1 and full.info.fused_step == 1 and full.info.patch_waves == 3
    # launches of this size (32768 items, or 16384 per group) run two waves per item, in the four-rows-per-group form (launch_chunk)
    assert full.info.patch_two_wave_min_items == 6144 and full.info.patch_big_min_items == 16384
    # ... and an arena of this size comes from the virtual-memory API in 1-GiB chunks at a 1-GiB-aligned address (a silent fall-back to the
    # allocator's memory would cost 15-20 % of the step rate with every numeric check below still green); the small exact engine stays on torch
    assert full.arena_kind == "vmm" and full.arena.chunk_bytes == 1 << 30 and full.arena.data_ptr() % (1 << 30) == 0, getattr(full, "arena_fallback_reason", None)
    assert exact.arena_kind in ("torch", "vmm")
    print(f"[{name}] arena {full.info.arena_bytes / 1e9:.1f} GB, {full.info.cov_slot_bytes / 1e6:.2f} MB of columns per env")
    white = torch.empty((B, N), dtype=torch.float32, device="cuda")
    full.normal_rows(white, N, 11, 1 << 40)
    noise = torch.empty((T, B, 9), dtype=torch.float32, device="cuda")
    full.normal_rows(noise, 9, 11, 2 << 40)
    acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(T)]
    init = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(B, 1)
    samples = [0, 1, S - 1, B // 2, B - 1]

    def episode(eng, n):
        eng.reset(white_noise=white[:n])
        prev, rewards, traces = init[:n], [], [torch.stack([eng.read_diag(e).double().sum() for e in samples if e < n])]
        for t in range(T):
            r, s = eng.step(acts[t][:n], prev, meas_noise=noise[t, :n])
            assert int(s.abs().sum()) == 0, (name, t)
            rewards.append(r.clone())
            traces.append(torch.stack([eng.read_diag(e).double().sum() for e in samples if e < n]))
            prev = acts[t][:n]
        return torch.stack(rewards), torch.stack(traces)

    r_full, tr_full = episode(full, B)
    r_exact, _ = episode(exact, S)
    # (1) invariants on all envs
    assert bool(torch.isfinite(r_full).all()) and bool((r_full >= 0).all())
    assert bool((tr_full[1:] < tr_full[:-1]).all())  # every step removes variance (trace monotone)
    ranks = host(full.ranks())
    assert ranks.min() >= T and ranks.max() <= 9 * T  # m in [1, 9] per step
    # (2) the first S envs against the exact factor mode (full columns, k_prepare + k_gain), same inputs
    worst = float((r_full[:, :S].double() - r_exact.double()).abs().max())
    print(f"[{name}] worst |reward(window 10, fused) - reward(exact)| over {S} envs x {T} steps = {worst:.2e}")
    assert worst < TOL
    assert torch.equal(full.ranks()[:S], exact.ranks())
    for e in (0, 1, S - 1):
        assert float((full.read_mean(e).double() - exact.read_mean(e).double()).abs().max()) < TOL
        assert float((full.read_diag(e).double() - exact.read_diag(e).double()).abs().max()) < TOL
    #     and env 0 against the fp64 factor-form oracle, step by step
    gt0 = host(full.read_gt(0)).reshape(grid, grid)
    assert np.max(np.abs(gt0 - orc.grf_from_white_noise(host(white[0]).reshape(grid, grid), 5.0))) < TOL
    fs = orc.factor_reset(ocfg)
    prev = np.array([2.0, 2.0, 14.0])
    for t in range(T):
        a = host(acts[t][0])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        z = orc.observe(ocfg, gt0, a, host(noise[t, 0, :m]))
        mask = orc.adaptive_mask(fs.mean, fs.diag, 0.4, 0.0)
        before = fs.diag.copy()
        orc.factor_step(ocfg, fs, a, z=z)
        want = orc.reward_from_diags(before, fs.diag, a, prev, UAV, mask)
        assert abs(float(r_full[t, 0]) - want) < TOL, (name, t, float(r_full[t, 0]), want)
        prev = a
    assert np.max(np.abs(host(full.read_mean(0)).ravel() - fs.mean)) < TOL
    assert np.max(np.abs(host(full.read_diag(0)).ravel() - fs.diag)) < TOL
    # cached diag == diag(P0 - U U^T) on a sampled env (densified on the device: N x N floats)
    e = B - 1
    P = full.read_cov(e)
    assert float((torch.diagonal(P).double() - full.read_diag(e).double()).abs().max()) < TOL
    assert float((P - P.T).abs().max()) < 1e-6
    del P
    # (3) reproducibility: the same episode again, bit for bit, on the fused kernel
    r_again, _ = episode(full, B)
    assert torch.equal(r_again, r_full)


def test_tree_steps_200x200_vs_chained_factor_oracle():
    """ipp_tree_step on the configs[4] grid: a 9-node tree down to depth 6 from a root with executed steps; every reward
    and sampled node diagonals against the fp64 factor-form oracle chained along the paths (a dense P is 12.8 GB here)."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    fresh_gpu()
    dim = 200
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    ocfg = orc.OracleConfig(x_dim=dim, y_dim=dim)
    eng = IPPEngine(cfg, capacity=2, state="factor", rank_cap=128, window_rows=-1, fixed_prior=True, node_capacity=16, max_batch=16)
    rs = np.random.RandomState(4)
    gt = rs.uniform(0.0, 1.0, size=(dim, dim))
    eng.reset(env_ids=[1], gt=gt[None])
    fs = orc.factor_reset(ocfg)
    prev = np.array([2.0, 2.0, 14.0])
    centre = np.array([120, 77])

    def random_action():
        c = np.clip(centre + rs.randint(-3, 4, size=2), 0, dim - 1)
        return np.array([4.0 * c[0] + 2.0, 4.0 * c[1] + 2.0, float(rs.choice([6.0, 8.0, 12.0, 14.0]))])

    for _ in range(4):  # the root: executed steps (the mean moves, so the adaptive mask is not trivial)
        a = random_action()
        eps = rs.normal(size=9)
        eng.step(a[None], prev[None], env_ids=[1], meas_noise=eps[None])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        orc.factor_step(ocfg, fs, a, z=orc.observe(ocfg, gt, a, eps[:m]))
        prev = a
    root_mean = fs.mean.copy()
    state_of, prev_of, path_of, depth_of = {None: fs}, {None: prev}, {None: []}, {None: 0}
    plan = [(0, None), (1, None), (2, 0), (3, 0), (4, 2), (5, 4), (6, 1), (7, 5), (8, 7)]  # node 8 sits at depth 6
    for nid, par in plan:
        depth_of[nid] = depth_of[par] + 1
    for level in range(1, 7):
        batch = [(nid, par) for nid, par in plan if depth_of[nid] == level]
        acts = np.array([random_action() for _ in batch])
        prevs = np.array([prev_of[par] for _, par in batch])
        reward, status = eng.tree_step([1] * len(batch), [pad(path_of[par]) for _, par in batch], acts, prevs,
                                       new_ids=[nid for nid, _ in batch])
        assert int(status.abs().sum()) == 0
        for k, (nid, par) in enumerate(batch):
            child = copy.deepcopy(state_of[par])
            mask = orc.adaptive_mask(root_mean, child.diag, 0.4, 0.0)  # the map mean is the root's (mcts.py:239 passes it on)
            before = child.diag.copy()
            orc.factor_step(ocfg, child, acts[k], z=None)
            want = orc.reward_from_diags(before, child.diag, acts[k], prevs[k], UAV, mask)
            assert abs(float(reward[k]) - want) < TOL, (level, nid, float(reward[k]), want)
            state_of[nid], prev_of[nid], path_of[nid] = child, acts[k], path_of[par] + [nid]
    for nid in (0, 3, 5, 8):
        assert np.max(np.abs(host(eng.tree_diag(nid)) - state_of[nid].diag)) < TOL
    # the root slot is untouched
    assert eng.rank(1) == fs.U.shape[1] and np.max(np.abs(host(eng.read_diag(1)) - fs.diag)) < TOL


def test_fullsize_tree_wave_1024_roots_256_sims_200x200():
    """configs[4] at full size: 1024 roots x 256 simulations x depth 5 on 200x200, 4 simulations per root per launch
    (4096 items per ipp_tree_step), node storage recycled between waves.  Invariants on every item, two sampled
    simulations against the factor-form oracle, and the whole search twice with identical bits."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.vec_env import cell_centre_actions

    fresh_gpu()
    grid, roots, sims, depth, wave, root_steps = 200, 1024, 256, 5, 4, 3
    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    ocfg = orc.OracleConfig(x_dim=grid, y_dim=grid)
    N, n_items = cfg.n_cells, roots * wave
    eng = IPPEngine(cfg, capacity=roots, state="factor", rank_cap=9 * (root_steps + depth + 1), window_rows=-1, fixed_prior=True,
                    node_capacity=n_items * depth, max_batch=n_items)
    white = torch.empty((roots, N), dtype=torch.float32, device="cuda")
    eng.normal_rows(white, N, 5, 1 << 40)
    eng.reset(white_noise=white)
    noise = torch.empty((root_steps, roots, 9), dtype=torch.float32, device="cuda")
    eng.normal_rows(noise, 9, 5, 2 << 40)
    prev = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(roots, 1)
    root_acts = []
    for t in range(root_steps):
        a = torch.as_tensor(cell_centre_actions(cfg, t, 0, roots, roots, ALTS), device="cuda")
        _, s = eng.step(a, prev, meas_noise=noise[t])
        assert int(s.abs().sum()) == 0
        root_acts.append(a)
        prev = a
    root_ids = torch.arange(roots, dtype=torch.int32, device="cuda").repeat_interleave(wave)
    prev_items = prev.repeat_interleave(wave, dim=0)
    paths = torch.full((depth, n_items, 6), -1, dtype=torch.int32, device="cuda")
    new_ids = [(d * n_items + torch.arange(n_items, device="cuda")).to(torch.int32) for d in range(depth)]
    for d in range(1, depth):
        paths[d] = paths[d - 1]
        paths[d, :, d - 1] = new_ids[d - 1]
    root_diag_sum = torch.stack([eng.read_diag(e).double().sum() for e in (0, roots - 1)])
    root_ranks = eng.ranks().clone()

    def search(keep_wave=None):
        """sims / wave waves; the actions of wave w, level d come from RandomState(10000 + 1000 + 8 w + d)."""
        out = []
        for w in range(sims // wave):
            p = prev_items
            for d in range(depth):
                a = torch.as_tensor(cell_centre_actions(cfg, 1000 + 8 * w + d, 0, n_items, n_items, ALTS), device="cuda")
                r, s = eng.tree_step(root_ids, paths[d], a, p, new_ids=new_ids[d])
                assert int(s.abs().sum()) == 0
                out.append(r.clone())
                p = a
            if keep_wave == w:
                return torch.stack(out)
        return torch.stack(out)

    r1 = search()
    assert r1.shape == (sims // wave * depth, n_items)
    assert bool(torch.isfinite(r1).all()) and bool((r1 >= 0).all())
    # a node's state has less variance than its parent's, level by level (last wave's nodes are still stored)
    for item in (0, n_items - 1):
        tr = [float(eng.tree_diag(d * n_items + item).double().sum()) for d in range(depth)]
        assert all(tr[d + 1] < tr[d] for d in range(depth - 1)) and tr[0] < float(root_diag_sum[0 if item == 0 else 1])
    # roots untouched by 256 x 1024 simulations
    assert torch.equal(eng.ranks(), root_ranks)
    assert torch.equal(torch.stack([eng.read_diag(e).double().sum() for e in (0, roots - 1)]), root_diag_sum)
    # two sampled simulations of the LAST wave against the factor-form oracle (root state rebuilt from the same inputs)
    last = sims // wave - 1
    for item in (1, n_items - 2):
        e = item // wave
        gt = host(eng.read_gt(e)).reshape(grid, grid)
        fs = orc.factor_reset(ocfg)
        pv = np.array([2.0, 2.0, 14.0])
        for t in range(root_steps):
            a = host(root_acts[t][e])
            m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
            orc.factor_step(ocfg, fs, a, z=orc.observe(ocfg, gt, a, host(noise[t, e, :m])))
            pv = a
        root_mean = fs.mean.copy()
        for d in range(depth):
            a = cell_centre_actions(cfg, 1000 + 8 * last + d, 0, n_items, n_items, ALTS)[item]
            mask = orc.adaptive_mask(root_mean, fs.diag, 0.4, 0.0)
            before = fs.diag.copy()
            orc.factor_step(ocfg, fs, a, z=None)
            want = orc.reward_from_diags(before, fs.diag, a, pv, UAV, mask)
            got = float(r1[last * depth + d, item])
            assert abs(got - want) < TOL, (item, d, got, want)
            pv = a
        assert np.max(np.abs(host(eng.tree_diag((depth - 1) * n_items + item)) - fs.diag)) < TOL
    # the whole search again: identical bits
    r2 = search()
    assert torch.equal(r1, r2)


def test_fullsize_search_through_the_driver_1024_roots_256_sims_200x200():
    """configs[4] through the real search driver (VectorMCTS: PUCT, valid-action mask, forced playouts, Dirichlet noise,
    transposition-aware backup; stub network): 1024 roots x 256 simulations on 200x200, 4 simulations in flight per root.
    Invariants on every root and every node; the root env slots stay untouched."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.planning.mcts_zero.vector_mcts import VectorMCTS
    from ipp_rl_amd.vec_env import cell_centre_actions

    fresh_gpu()
    grid, roots, sims, W, horizon, root_steps = 200, 1024, 256, 4, 5, 3
    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    eng = IPPEngine(cfg, capacity=roots, state="factor", rank_cap=9 * (root_steps + horizon + 2), window_rows=-1, fixed_prior=True,
                    node_capacity=roots * (sims + W), max_batch=roots * W)
    white = torch.empty((roots, cfg.n_cells), dtype=torch.float32, device="cuda")
    eng.normal_rows(white, cfg.n_cells, 9, 1 << 40)
    eng.reset(white_noise=white)
    noise = torch.empty((root_steps, roots, 9), dtype=torch.float32, device="cuda")
    eng.normal_rows(noise, 9, 9, 2 << 40)
    prev = np.tile([2.0, 2.0, 14.0], (roots, 1))
    for t in range(root_steps):
        a = cell_centre_actions(cfg, t, 0, roots, roots, [8.0, 14.0])
        _, s = eng.step(a, prev, meas_noise=noise[t])
        assert int(s.abs().sum()) == 0
        prev = a
    ranks0 = eng.ranks().clone()
    diag0 = eng.read_diag(roots - 1).clone()
    hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
                 dirichlet_alpha=1.0, dirichlet_eps=0.25, num_mcts_simulations=sims)
    meta = {"budget": 100.0, "initial_budget": 100.0, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
            "altitude_spacing": 6.0, "uav_specifications": UAV, "scenario_info": {"value_threshold": 0.4, "interval_factor": 0}}
    mcts = VectorMCTS(eng, hyper, meta, lambda reqs: [(None, 0.3)] * len(reqs), sims_in_flight=W, seed=1)
    out = mcts.get_policy(list(range(roots)), prev, [100.0] * roots)
    n = mcts.n_count
    print(f"[configs[4] driver] {mcts.stats['nodes']} nodes, {mcts.stats['device_steps']} device steps in {mcts.stats['launches']} launches")
    for j in range(roots):
        rt = int(mcts.root_ids[j])
        K = int(mcts.n_K[rt])
        assert mcts.n_Ns[rt] == sims - W == mcts.t_Nsa[rt, :K].sum()  # the first wave of W simulations expands the root
        policy, valid = out[j]
        assert abs(sum(policy.values()) - 1.0) < 1e-9 and set(policy) <= set(int(i) for i in valid)
    assert np.all(mcts.t_Nsa[:n].sum(axis=1) == mcts.n_Ns[:n]) and np.all(mcts.t_Nsa[:n] >= 0)  # virtual visits undone
    num = mcts.t_num[:n]
    assert not np.any(np.isinf(num)) and np.all(num[~np.isnan(num)] > 0)  # every traversed edge has its (positive) trace reduction
    q = mcts.t_Qsa[:n]
    assert np.all(np.isfinite(q)) and np.all(q >= 0)
    assert mcts.stats["launches"] <= (sims // W) * (horizon + 1)  # one launch per level and wave, shared by all roots
    assert torch.equal(eng.ranks(), ranks0) and torch.equal(eng.read_diag(roots - 1), diag0)


def test_fullsize_device_search_1024_roots_256_sims_200x200():
    """configs[4] through the DEVICE-side search (DeviceMCTS, csrc/k_mcts.h): 1024 roots x 256 simulations on 200x200, 4 in
    flight per root, Dirichlet noise and random tie-breaking from the counter-based device streams.  Invariants over all
    node tables read back from the device; two searches with one seed are bit-identical; root env slots untouched."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.planning.mcts_zero.device_mcts import DeviceMCTS
    from ipp_rl_amd.vec_env import cell_centre_actions

    fresh_gpu()
    grid, roots, sims, W, horizon, root_steps = 200, 1024, 256, 4, 5, 3
    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    eng = IPPEngine(cfg, capacity=roots, state="factor", rank_cap=9 * (root_steps + horizon + 2), window_rows=-1, fixed_prior=True,
                    node_capacity=roots * (sims + W), max_batch=roots * W)
    white = torch.empty((roots, cfg.n_cells), dtype=torch.float32, device="cuda")
    eng.normal_rows(white, cfg.n_cells, 9, 1 << 40)
    eng.reset(white_noise=white)
    noise = torch.empty((root_steps, roots, 9), dtype=torch.float32, device="cuda")
    eng.normal_rows(noise, 9, 9, 2 << 40)
    prev = np.tile([2.0, 2.0, 14.0], (roots, 1))
    for t in range(root_steps):
        a = cell_centre_actions(cfg, t, 0, roots, roots, [8.0, 14.0])
        _, s = eng.step(a, prev, meas_noise=noise[t])
        assert int(s.abs().sum()) == 0
        prev = a
    ranks0 = eng.ranks().clone()
    diag0 = eng.read_diag(roots - 1).clone()
    hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
                 dirichlet_alpha=1.0, dirichlet_eps=0.25, num_mcts_simulations=sims)
    meta = {"budget": 100.0, "initial_budget": 100.0, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
            "altitude_spacing": 6.0, "uav_specifications": UAV, "scenario_info": {"value_threshold": 0.4, "interval_factor": 0}}
    mcts = DeviceMCTS(eng, hyper, meta, None, sims_in_flight=W, tie_break="random", seed=1, leaf_value=0.3)
    out = mcts.get_policy(list(range(roots)), prev, [100.0] * roots)
    pieces = mcts._subs_used or [mcts]  # (the default search runs as two groups of roots with their own tables)
    assert len(pieces) == mcts.groups == 2
    npr = mcts.nodes_per_root
    used_max = 0
    for piece in pieces:
        b = piece._buf
        used = b["root_count"].cpu().numpy()
        used_max = max(used_max, int(used.max()))
        assert used.max() < npr and int(b["dev_count"].max()) <= mcts.dev_per_root and int(b["err"].abs().sum()) == 0
        live = (torch.arange(npr, device="cuda")[None, :] < b["root_count"][:, None].to(torch.int64)).reshape(-1)  # node ids in use
        exp = ((b["n_flags"] & 1) != 0) & live
        K = b["n_k"].to(torch.int64)
        col = torch.arange(b["t_idx"].shape[1], device="cuda")[None, :]
        inside = (col < K[:, None]) & exp[:, None]
        nsa, q, num, idx = b["t_nsa"], b["t_qsa"], b["t_num"], b["t_idx"]
        assert bool(((idx >= 0) == (col < K[:, None]))[exp].all())                                   # valid sets are exactly K long
        assert bool((idx[:, 1:] > idx[:, :-1])[inside[:, 1:]].all())                                 # ... and ascending
        assert bool(torch.equal(torch.where(inside, nsa, torch.zeros_like(nsa)).sum(dim=1)[exp], b["n_ns"][exp]))  # virtual visits undone
        assert bool((nsa[inside] >= 0).all()) and bool(torch.isfinite(q[inside]).all()) and bool((q[inside] >= 0).all())
        trav = inside & ~torch.isnan(num)
        assert bool(torch.isfinite(num[trav]).all()) and bool((num[trav] > 0).all())                 # every traversed edge has its trace reduction
        assert bool(((nsa > 0) & inside & torch.isnan(num)).sum() == 0)                              # no visited edge without one
    print(f"[configs[4] device search] {mcts.stats['nodes']} nodes (most per root {used_max} of {npr}), "
          f"{mcts.stats['device_steps']} device steps in {mcts.stats['launches']} launches")
    for j in range(roots):
        Kj = int(mcts.n_K[j])
        assert mcts.n_Ns[j] == sims - W == mcts.t_Nsa[j, :Kj].sum()
        policy, valid = out[j]
        assert abs(sum(policy.values()) - 1.0) < 1e-9 and set(policy) <= set(int(i) for i in valid)
    assert mcts.stats["launches"] <= (sims // W) * (horizon + 1)
    assert torch.equal(eng.ranks(), ranks0) and torch.equal(eng.read_diag(roots - 1), diag0)
    nsa1 = mcts.t_Nsa.copy()
    out2 = mcts.get_policy(list(range(roots)), prev, [100.0] * roots)
    assert np.array_equal(nsa1, mcts.t_Nsa) and all(out[j][0] == out2[j][0] for j in range(roots))


@pytest.mark.parametrize("grid,capacity,T,node_capacity", [
    (50, 64, 40, 0),        # configs[0] / [1]: 50x50
    (50, 4096, 40, 0),      # configs[1]
    (100, 256, 16, 0),      # configs[2] geometry (the full batch: test above)
    (50, 16384, 40, 0),     # configs[3] per-GPU share
    (200, 64, 11, 256),     # configs[4]: 200x200 roots + tree nodes
])
def test_baseline_configs_take_the_patch_kernels(grid, capacity, T, node_capacity):
    """Kernel selection per BASELINE config: the window of the fixed example prior puts every one of them on the patch layout."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    fresh_gpu()
    eng = IPPEngine(EngineConfig(x_dim=grid, y_dim=grid), capacity=capacity, state="factor", rank_cap=9 * T, window_rows=-1,
                    fixed_prior=True, node_capacity=node_capacity)
    assert eng.info.window_rows == 10
    assert eng.info.patch_layout == 1 and eng.info.fused_step == 1 and eng.info.patch_waves == 3
    assert eng.info.patch_two_wave_min_items == 6144 and eng.info.patch_big_min_items == 16384
    eng.close()


def test_two_wave_engines_keep_their_large_launch_instantiation(monkeypatch):
    """IPP_PATCH_WAVES=2 (the round-3 configuration, kept for A/B): launches of >= 16384 items take k_step_patch<2, 4, 6>."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    fresh_gpu()
    monkeypatch.setenv("IPP_PATCH_WAVES", "2")
    eng = IPPEngine(EngineConfig(x_dim=50, y_dim=50), capacity=64, state="factor", rank_cap=360, window_rows=-1, fixed_prior=True)
    assert eng.info.patch_waves == 2 and eng.info.patch_big_min_items == 16384 and eng.info.patch_two_wave_min_items == 0
    eng.close()
