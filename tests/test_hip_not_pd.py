"""
GPU test of the factor state's refusal of a non-positive-definite S (IPP_STATUS_NOT_PD = 2; mapping/mappings.py:200-215 is
the reference's inverse fallback, which the factor form cannot represent -- DESIGN.md section 4, INTEGRATION.md divergences):
the status and a NaN reward come back for THAT env only, its state is left as it was (no columns appended), and the other envs
of the same launch are bit-identical to a launch without it -- on the patch kernel (the batched driver's path) and on the
band-tile kernels.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALTS = [float(a) for a in range(5, 15)]


@pytest.mark.parametrize("patch", [1, 0])
def test_not_pd_env_is_reported_and_isolated(patch):
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.vec_env import cell_centre_actions

    old = os.environ.get("IPP_PATCH")
    os.environ["IPP_PATCH"] = str(patch)
    try:
        cfg = EngineConfig(x_dim=50, y_dim=50)
        B, bad = 12, 5
        engines = [IPPEngine(cfg, capacity=B, state="factor", rank_cap=90, window_rows=-1, fixed_prior=True) for _ in range(2)]
    finally:
        if old is None:
            os.environ.pop("IPP_PATCH", None)
        else:
            os.environ["IPP_PATCH"] = old
    assert engines[0].info.patch_layout == patch
    rs = np.random.RandomState(3)
    gt = torch.as_tensor(rs.uniform(size=(B, 2500)), dtype=torch.float32, device="cuda")
    for e in engines:
        e.reset(gt=gt)
    # engine 1: env `bad` gets a prior the column window was not sized for -> its planes are NaN -> S is not positive definite
    too_long = np.array([[cfg.signal_variance, cfg.length_scale * 1.1]])
    engines[1].reset(env_ids=[bad], prior_scale=too_long, gt=gt[bad:bad + 1])
    prev = [torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(B, 1) for _ in range(2)]
    for t in range(6):
        a = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), dtype=torch.float64, device="cuda")
        noise = torch.as_tensor(rs.normal(size=(B, 9)), dtype=torch.float32, device="cuda")
        out = []
        for e, p in zip(engines, prev):
            r, s = e.step(a, p, meas_noise=noise)
            out.append((r.clone(), s.clone()))
        (r0, s0), (r1, s1) = out
        assert int(s0.abs().sum()) == 0
        assert int(s1[bad]) == 2 and bool(torch.isnan(r1[bad]))           # IPP_STATUS_NOT_PD, reward NaN
        others = [i for i in range(B) if i != bad]
        assert int(s1[others].abs().sum()) == 0
        assert torch.equal(r0[others], r1[others])                          # neighbours of the launch: bit-identical
        prev = [a, a.clone()]
    assert int(engines[1].rank(bad)) == 0                                   # the refused steps appended nothing
    for i in (0, bad - 1, bad + 1, B - 1):
        assert torch.equal(engines[0].read_mean(i), engines[1].read_mean(i))
        assert torch.equal(engines[0].read_diag(i), engines[1].read_diag(i))
        assert int(engines[0].rank(i)) == int(engines[1].rank(i))


@pytest.mark.parametrize("cap", [8, 40])
def test_column_records_that_overflow_the_lds_staging(cap, monkeypatch):
    """k_step_patch / k_tree_patch keep one record per contributing column in LDS (View::pcap, ~200 at the headline); the rest
    goes to the item's global scratch block and is served by the generic request group.  With the capacity forced down to 8 / 40
    records (IPP_PATCH_CAP) most columns of a 30-step episode of clustered revisits overflow: rewards, means, variances, dense
    covariances and chained tree steps must equal the default engine's bit for bit."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine

    cfg = EngineConfig(x_dim=50, y_dim=50)
    B, T = 6, 30
    rs = np.random.RandomState(4)
    gt = torch.as_tensor(rs.uniform(size=(B, 2500)), dtype=torch.float32, device="cuda")
    centre = rs.randint(10, 40, size=(B, 2))
    acts = []
    for t in range(T):  # revisits within two cells of a centre: nearly every stored column reaches every footprint
        cell = np.clip(centre + rs.randint(-2, 3, size=(B, 2)), 0, 49)
        acts.append(np.stack([4.0 * cell[:, 0] + 2.0, 4.0 * cell[:, 1] + 2.0, rs.choice(ALTS, B)], axis=1))
    noise = rs.normal(size=(T, B, 9))
    outs = []
    for forced in (None, cap):
        if forced is None:
            monkeypatch.delenv("IPP_PATCH_CAP", raising=False)
        else:
            monkeypatch.setenv("IPP_PATCH_CAP", str(forced))
        eng = IPPEngine(cfg, capacity=B, state="factor", rank_cap=9 * (T + 4), window_rows=-1, fixed_prior=True, node_capacity=3 * B,
                        max_batch=64)
        assert eng.info.patch_layout == 1
        eng.reset(gt=gt)
        prev = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(B, 1)
        rewards = []
        for t in range(T):
            a = torch.as_tensor(acts[t], dtype=torch.float64, device="cuda")
            r, s = eng.step(a, prev, meas_noise=torch.as_tensor(noise[t], dtype=torch.float32, device="cuda"))
            assert int(s.abs().sum()) == 0
            rewards.append(r.clone())
            prev = a
        roots = torch.arange(B, dtype=torch.int32, device="cuda")
        path = torch.full((B, 6), -1, dtype=torch.int32, device="cuda")
        tree = []
        for d in range(3):  # chained tree steps from the deep states (root columns + node columns in the records)
            a = prev.clone()
            a[:, 0] = torch.clamp(a[:, 0] + 4.0 * (d + 1), 2.0, 198.0)
            a[:, 2] = ALTS[(3 * d + 2) % 10]
            ids = torch.arange(d * B, (d + 1) * B, dtype=torch.int32, device="cuda")
            r, s = eng.tree_step(roots, path, a, prev, ids)
            assert int(s.abs().sum()) == 0
            path[:, d] = ids
            prev = a
            tree += [r.clone(), eng.tree_diag(int(ids[0])).clone()]
        outs.append(dict(rewards=torch.stack(rewards), mean=torch.stack([eng.read_mean(e) for e in range(B)]),
                         diag=torch.stack([eng.read_diag(e) for e in range(B)]), cov=eng.read_cov(1).clone(), tree=tree,
                         ranks=eng.ranks().clone()))
        eng.close()
    a, b = outs
    assert int(a["ranks"].max()) > 100
    for key in ("rewards", "mean", "diag", "cov", "ranks"):
        assert torch.equal(a[key], b[key]), key
    for x, y in zip(a["tree"], b["tree"]):
        assert torch.equal(x, y)


def _revisit_actions(rs, kind, B, t, dim=50):
    """Worst cases for S = H P H^T + R on the factor state: the SAME footprint measured again and again -- H P H^T shrinks like R / k
    while it is evaluated as P0 - sum of squared fp32 column entries, so its rounding error is largest relative to what is left."""
    if kind == "1x1@5m":      # one cell, R = nv(5 m) = 0.0316
        cell = np.tile(np.array([[17, 23]]), (B, 1))
        alt = np.full(B, 5.0)
    elif kind == "5x5@14m":   # 5 x 5 cells -> 3 x 3 blocks of rf = 2 (partial border blocks), R = 8 nv(14 m)
        cell = np.tile(np.array([[30, 12]]), (B, 1))
        alt = np.full(B, 14.0)
    elif kind == "corner@14m":  # clipped at the grid corner: 3 x 3 cells, 2 x 2 blocks
        cell = np.tile(np.array([[0, 49]]), (B, 1))
        alt = np.full(B, 14.0)
    else:                      # jitter by one cell and by altitude: overlapping footprints of every shape
        cell = np.array([[25, 25]]) + rs.randint(-1, 2, size=(B, 2))
        alt = rs.choice(ALTS, B)
    return np.stack([4.0 * cell[:, 0] + 2.0, 4.0 * cell[:, 1] + 2.0, alt], axis=1)


@pytest.mark.parametrize("kind", ["1x1@5m", "5x5@14m", "corner@14m", "jitter"])
@pytest.mark.parametrize("split", [0, 1])
def test_S_stays_positive_definite_on_the_worst_reachable_states(kind, split, monkeypatch):
    """mapping/mappings.py:200-215 (the inverse fallback behind a failed Cholesky) against the factor engine's refusal (status 2):
    the refusal is UNREACHABLE on states the engine itself produces.  S = H P H^T + R with P = P0 - U U^T positive semi-definite in
    exact arithmetic, so lambda_min(S) >= R; in the engine H P H^T comes from fp32 columns summed in fp64, whose error after 40
    revisits of one footprint is ~1e-6 -- four orders of magnitude below R (0.03 at 5 m, 0.39 at 14 m).  Driven here: 40 steps (a
    whole episode, the configured rank cap) on the headline configuration (fp32 patches, window 10, fixed prior), every step's S read
    back (ipp_debug_step_item): status 0, lambda_min(S) >= R / 2 -- in fact >= R (1 - 1e-4) -- and the oracle's S within 1e-5."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from oracle import ipp_oracle as orc

    monkeypatch.setenv("IPP_SPLIT", str(split))
    cfg = EngineConfig(x_dim=50, y_dim=50)
    B, T = 4, 40
    eng = IPPEngine(cfg, capacity=B, state="factor", rank_cap=9 * T, window_rows=-1, fixed_prior=True)
    monkeypatch.delenv("IPP_SPLIT")
    assert eng.info.patch_layout == 1 and int(eng.info.window_rows) == 10 and int(eng.info.patch_split_min_items) == split
    ocfg = orc.OracleConfig(x_dim=50, y_dim=50)
    rs = np.random.RandomState(11)
    white = rs.normal(size=(B, 50, 50))
    eng.reset(white_noise=white)
    ost = orc.env_reset(ocfg, white[0])
    eng.debug_capture(True)
    prev = np.tile(np.array([2.0, 2.0, 14.0]), (B, 1))
    worst_ratio, worst_S = np.inf, 0.0
    for t in range(T):
        acts = _revisit_actions(rs, kind, B, t)
        eps = rs.normal(size=(B, 9))
        reward, status = eng.step(acts, prev, meas_noise=eps)
        torch.cuda.synchronize()
        assert int(status.abs().sum()) == 0, (kind, t, status.tolist())
        assert bool(torch.isfinite(reward).all())
        for b in range(B):
            it = eng.debug_item(b)
            R = it["rf"] ** 3 * it["noise_var"]
            lam = float(np.linalg.eigvalsh(it["S"]).min())
            worst_ratio = min(worst_ratio, lam / R)
            assert lam >= 0.5 * R, (kind, t, b, lam, R)
        # env 0 against the fp64 oracle (dense P): the same S
        m = orc.num_measurements(orc.project_fov(ocfg, acts[0]), orc.resolution_factor(acts[0]))
        out = orc.env_step(ocfg, ost, acts[0], eps[0, :m])
        worst_S = max(worst_S, float(np.max(np.abs(eng.debug_item(0)["S"] - out["terms"].S))))
        assert abs(float(reward[0]) - out["reward"]) < 1e-5
        prev = acts
    assert worst_ratio > 1.0 - 1e-4, (kind, worst_ratio)
    assert worst_S < 1e-5
    assert int(eng.ranks().max()) <= 9 * T
    eng.close()
