"""
GPU edge cases and size-independent properties of the HIP step engine (through the C-ABI):
border / clipped footprints, every altitude level 4..30 m (m up to 25 -> the second compiled instantiation),
non-square grids, status codes, out-of-place steps, forks, predict-only scoring, full-size (cfg2) invariants and
sharding equivalence.  Oracle = oracle/ipp_oracle.py (fp64, pinned against the reference's golden vectors).
"""
import numpy as np
import pytest

from oracle import ipp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def make(dim_x, dim_y=None, state="factor", capacity=2, res=4.0, **kw):
    from ipp_rl_amd import EngineConfig, IPPEngine

    dim_y = dim_y or dim_x
    cfg = EngineConfig(x_dim=dim_x, y_dim=dim_y, resolution=res, **kw.pop("cfg", {}))
    eng = IPPEngine(cfg, capacity=capacity, state=state, rank_cap=kw.pop("rank_cap", 512), **kw)
    eng.debug_capture()  # (S / z of every step for debug_item)
    ocfg = orc.OracleConfig(x_dim=dim_x, y_dim=dim_y, resolution=res, coeff_a=cfg.coeff_a, coeff_b=cfg.coeff_b)
    return eng, ocfg


@pytest.mark.parametrize("state", ["dense", "factor"])
def test_all_altitudes_and_borders_m_up_to_25(state):
    """alt 4..30 at res 4 (footprints 1x1 .. 9x9, rf 1 and 2, m <= 25) incl. corners / edges / pos == W*res."""
    dim = 24
    eng, ocfg = make(dim, state=state, max_measurements=25)
    assert eng.meas_cap == 25
    rs = np.random.RandomState(5)
    gt = rs.uniform(size=(dim, dim))
    eng.reset(env_ids=[0], gt=gt[None])
    st = orc.EnvState(mean=0.5 * np.ones((dim, dim)), P=orc.matern_prior(ocfg), gt=gt)
    W = dim * 4.0
    spots = [(2.0, 2.0), (W - 2, W - 2), (2.0, W - 2), (W, W), (0.0, 50.0), (50.0, 0.0), (W / 2, W / 2), (33.3, 71.9)]
    prev = np.array([2.0, 2.0, 14.0])
    for t, alt in enumerate(range(4, 31)):
        x, y = spots[t % len(spots)]
        a = np.array([x, y, float(alt)])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        eps = rs.normal(size=25)
        reward, status = eng.step(a[None], prev[None], env_ids=[0], meas_noise=eps[None])
        it = eng.debug_item(0)
        out = orc.env_step(ocfg, st, a, eps[:m])
        assert int(status[0]) == 0 and it["m"] == m and it["fov"] == orc.project_fov(ocfg, a)
        assert abs(float(reward[0]) - out["reward"]) < TOL
        assert np.max(np.abs(it["z"] - out["z"].ravel())) < 1e-6
        prev = a
    assert np.max(np.abs(host(eng.read_mean(0)) - st.mean)) < TOL
    assert np.max(np.abs(host(eng.read_cov(0)) - st.P)) < TOL


@pytest.mark.parametrize("state", ["dense", "factor"])
def test_resolution_one_rf1_5x5(state):
    """res = 1 m/cell: a 5x5 footprint at rf = 1 gives m = 25 (res in {1, 4})."""
    dim = 16
    eng, ocfg = make(dim, state=state, res=1.0, max_measurements=25)
    rs = np.random.RandomState(9)
    gt = rs.uniform(size=(dim, dim))
    eng.reset(env_ids=[0], gt=gt[None])
    st = orc.EnvState(mean=0.5 * np.ones((dim, dim)), P=orc.matern_prior(ocfg), gt=gt, prev=np.array([0.5, 0.5, 5.0]))
    prev = st.prev.copy()
    for alt, (x, y) in [(5.0, (8.5, 8.5)), (5.0, (0.5, 15.5)), (4.0, (3.5, 9.5)), (3.0, (12.5, 2.5))]:
        a = np.array([x, y, alt])
        m = orc.num_measurements(orc.project_fov(ocfg, a), 1)
        eps = rs.normal(size=25)
        reward, status = eng.step(a[None], prev[None], env_ids=[0], meas_noise=eps[None])
        out = orc.env_step(ocfg, st, a, eps[:m])
        assert int(status[0]) == 0 and abs(float(reward[0]) - out["reward"]) < TOL
        prev = a
    assert np.max(np.abs(host(eng.read_cov(0)) - st.P)) < TOL and np.max(np.abs(host(eng.read_mean(0)) - st.mean)) < TOL


@pytest.mark.parametrize("state", ["dense", "factor"])
def test_non_square_grid(state):
    """flat index = x_dim * row + col on a 12 x 10 grid (ground truth supplied: the reference's GRF is square-only)."""
    eng, ocfg = make(10, 12, state=state)
    rs = np.random.RandomState(2)
    gt = rs.uniform(size=(12, 10))
    eng.reset(env_ids=[0], gt=gt[None])
    st = orc.EnvState(mean=0.5 * np.ones((12, 10)), P=orc.matern_prior(ocfg), gt=gt)
    prev = np.array([2.0, 2.0, 14.0])
    for _ in range(12):
        a = np.array([4.0 * rs.randint(0, 10) + 2, 4.0 * rs.randint(0, 12) + 2, float(rs.randint(5, 15))])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        eps = rs.normal(size=9)
        reward, _ = eng.step(a[None], prev[None], env_ids=[0], meas_noise=eps[None])
        out = orc.env_step(ocfg, st, a, eps[:m])
        assert abs(float(reward[0]) - out["reward"]) < TOL
        prev = a
    assert np.max(np.abs(host(eng.read_cov(0)) - st.P)) < TOL
    assert np.max(np.abs(host(eng.read_mean(0)) - st.mean)) < TOL


def test_status_codes():
    from ipp_rl_amd import _ffi

    # m > compiled cap (9): alt 25 m -> 7x7 rf 2 -> m = 16
    eng, _ = make(20, state="factor")
    eng.reset(env_ids=[0])
    r, s = eng.step(np.array([[42.0, 42.0, 25.0]]), np.array([[2.0, 2.0, 14.0]]), env_ids=[0])
    assert int(s[0]) == _ffi.STATUS_BAD_FOOTPRINT and float(r[0]) == 0.0 and eng.rank(0) == 0
    # non-finite position
    r, s = eng.step(np.array([[np.nan, 42.0, 8.0]]), np.array([[2.0, 2.0, 14.0]]), env_ids=[0])
    assert int(s[0]) == _ffi.STATUS_BAD_FOOTPRINT
    # rank cap: 2 steps of m = 9 fit into rank_cap 20, the third does not: reward still valid, state untouched
    eng, ocfg = make(20, state="factor", rank_cap=20)
    eng.reset(env_ids=[0])
    prev = np.array([[2.0, 2.0, 14.0]])
    for k, a in enumerate([[42.0, 42.0, 8.0], [10.0, 50.0, 9.0], [62.0, 22.0, 8.0]]):
        a = np.array([a])
        before = host(eng.read_diag(0))
        r, s = eng.step(a, prev, env_ids=[0], cov_only=True)
        if k < 2:
            assert int(s[0]) == 0
        else:
            assert int(s[0]) == _ffi.STATUS_RANK_FULL and eng.rank(0) == 18 and float(r[0]) > 0
            assert np.array_equal(host(eng.read_diag(0)), before)
        prev = a


def test_not_positive_definite_paths():
    """Negative sensor noise makes S indefinite: dense takes the reference's inverse fallback (mappings.py:200-215),
    the factor form refuses the step."""
    from ipp_rl_amd import _ffi

    neg = {"coeff_a": -3.0}
    dense, ocfg = make(8, state="dense", cfg=neg)
    factor, _ = make(8, state="factor", cfg=neg)
    ocfg.coeff_a = -3.0
    gt = np.linspace(0, 1, 64).reshape(8, 8)
    a, prev = np.array([[14.0, 14.0, 8.0]]), np.array([[2.0, 2.0, 14.0]])
    z = np.full((1, 9), 0.6)
    for eng in (dense, factor):
        eng.reset(env_ids=[0], gt=gt[None])
    r, s = dense.step(a, prev, env_ids=[0], meas_noise=z, adaptive=False, given_observation=True)
    assert int(s[0]) == _ffi.STATUS_CHOL_FALLBACK
    P0 = orc.matern_prior(ocfg)
    x, Pn, terms = orc.update_grid_map(ocfg, P0, 0.5 * np.ones((8, 8)), a[0], z[0])
    assert terms.used_fallback
    assert np.max(np.abs(host(dense.read_cov(0)) - Pn)) < 1e-4 * max(1.0, np.abs(Pn).max())
    assert np.max(np.abs(host(dense.read_mean(0)) - x)) < 1e-4 * max(1.0, np.abs(x).max())
    r, s = factor.step(a, prev, env_ids=[0], meas_noise=z, adaptive=False, given_observation=True)
    assert int(s[0]) == _ffi.STATUS_NOT_PD and np.isnan(float(r[0])) and factor.rank(0) == 0


@pytest.mark.parametrize("state,window_rows", [("dense", 0), ("factor", 0), ("factor", 12)])
def test_out_of_place_fork_and_predict_only(state, window_rows):
    dim = 14
    eng, ocfg = make(dim, state=state, capacity=6, window_rows=window_rows)
    rs = np.random.RandomState(3)
    eng.reset(white_noise=rs.normal(size=(6, dim, dim)))
    prev = np.tile([2.0, 2.0, 14.0], (6, 1))
    acts = np.stack([4.0 * rs.randint(0, dim, 6) + 2, 4.0 * rs.randint(0, dim, 6) + 2, rs.randint(5, 15, 6) * 1.0], 1)
    eng.step(acts, prev, meas_noise=rs.normal(size=(6, 9)))
    P0, m0, d0 = host(eng.read_cov(0)), host(eng.read_mean(0)), host(eng.read_diag(0))
    # predict-only scoring of 40 candidates from slot 0 (env id repeated): nothing may change, bit for bit
    cands = np.stack([4.0 * rs.randint(0, dim, 40) + 2, 4.0 * rs.randint(0, dim, 40) + 2, rs.randint(5, 15, 40) * 1.0], 1)
    eng2 = None
    r_pred, s_pred = eng.step(cands[:6], np.tile(acts[0], (6, 1)), env_ids=np.zeros(6, np.int32), predict_only=True,
                              cov_only=True)
    assert np.array_equal(host(eng.read_cov(0)), P0) and np.array_equal(host(eng.read_diag(0)), d0)
    # out-of-place: slot 0 -> slot 4 with the first candidate; source untouched, destination == in-place result on a fork
    eng.fork([0], [5])
    a = cands[:1]
    r_oop, _ = eng.step(a, acts[:1], env_ids=[0], dst_ids=[4], cov_only=True)
    r_inp, _ = eng.step(a, acts[:1], env_ids=[5], cov_only=True)
    assert np.array_equal(host(eng.read_cov(0)), P0) and np.array_equal(host(eng.read_mean(0)), m0)
    assert abs(float(r_oop[0]) - float(r_pred[0])) < 1e-6 and abs(float(r_oop[0]) - float(r_inp[0])) < 1e-6
    assert np.max(np.abs(host(eng.read_cov(4)) - host(eng.read_cov(5)))) < 1e-6
    assert np.array_equal(host(eng.read_gt(4)), host(eng.read_gt(0)))
    if state == "factor":
        assert eng.rank(4) == eng.rank(0) + eng.debug_item(0)["m"]
    del eng2


@pytest.mark.parametrize("window_rows", [0, -1])
def test_fullsize_invariants_and_sharding_equivalence(window_rows):
    """cfg2 size (4096 envs, 50x50, factor state): trace monotone, reward >= 0, cached diag == diag(P0 - U U^T) on
    sampled envs, and a 2-way shard (2 engines x 2048 envs) reproduces every env's result bit for bit -- on the exact
    path (k_prepare + k_gain) and on the bench's path (window from the prior, fused k_step_factor)."""
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions
    import torch

    B, dim, steps = 4096, 50, 6
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    alts = [float(a) for a in range(5, 15)]
    kw = dict(window_rows=window_rows, fixed_prior=window_rows < 0)
    full, _ = make(dim, state="factor", capacity=B, rank_cap=64, **kw)
    halves = [make(dim, state="factor", capacity=B // 2, rank_cap=64, **kw)[0] for _ in range(2)]
    assert full.info.window_rows == (10 if window_rows < 0 else 0)
    white = full.normal(B * dim * dim, seed=7).reshape(B, -1)
    full.reset(white_noise=white)
    for h, eng in enumerate(halves):
        eng.reset(white_noise=white[h * B // 2:(h + 1) * B // 2])
    prev = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(B, 1)
    trace_before = full.read_diag(0).sum() * 0 + torch.stack([full.read_diag(e).sum() for e in (0, 1, 2047, 2048, 4095)])
    for t in range(steps):
        acts = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts), device="cuda")
        eps = full.normal(B * 9, seed=100 + t).reshape(B, 9)
        r_full, s_full = full.step(acts, prev, meas_noise=eps)
        r_full = r_full.clone()
        assert int(s_full.abs().sum()) == 0 and bool((r_full >= 0).all())
        for h, eng in enumerate(halves):
            sl = slice(h * B // 2, (h + 1) * B // 2)
            r_h, _ = eng.step(acts[sl], prev[sl], meas_noise=eps[sl])
            assert torch.equal(r_h, r_full[sl])  # bit-identical regardless of how envs are sharded
        prev = acts
    for e in (0, 1, 2047, 2048, 4095):
        h, le = divmod(e, B // 2)
        assert torch.equal(full.read_mean(e), halves[h].read_mean(le))
        assert torch.equal(full.read_diag(e), halves[h].read_diag(le))
    trace_after = torch.stack([full.read_diag(e).sum() for e in (0, 1, 2047, 2048, 4095)])
    assert bool((trace_after < trace_before).all())
    P = host(full.read_cov(4095))
    assert np.max(np.abs(np.diag(P) - host(full.read_diag(4095)))) < TOL
    assert np.max(np.abs(P - P.T)) < 1e-6 and np.linalg.eigvalsh(P).min() > -1e-4


def test_vec_env_staged_ground_truth_matches_direct_reset():
    """VecIPPEnv's side-stream GRF staging (ipp_generate_grf -> ipp_reset(gt)) equals the direct white-noise reset."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    dim, B = 20, 8
    eng = IPPEngine(EngineConfig(x_dim=dim, y_dim=dim), capacity=B, state="factor", rank_cap=32)
    white = eng.normal(B * dim * dim, seed=11, subsequence=3).reshape(B, -1)
    eng.reset(white_noise=white)
    direct = [eng.read_gt(e).clone() for e in range(B)]
    staged = eng.generate_grf(white)
    eng.reset(gt=staged)
    for e in range(B):
        assert np.array_equal(host(eng.read_gt(e)), host(direct[e]))
    gt = host(direct[0])
    assert gt.min() == 0.0 and abs(gt.max() - 1.0) < 1e-6
    ref = orc.grf_from_white_noise(host(white[0]).reshape(dim, dim), 5.0)
    assert np.max(np.abs(gt - ref)) < TOL


def test_vec_env_staggered_schedule_and_determinism():
    """Staggered VecIPPEnv loop (bench driver): scheduled resets land on the right envs at the right steps, the
    side-stream ground truths staged one step ahead are the ones a direct reset would draw, and two instances
    with the same seed agree bit for bit (no dependence on stream timing)."""
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions
    import torch

    dim, B, T, steps = 20, 24, 6, 15
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    alts = [6.0, 8.0, 12.0]
    envs = [VecIPPEnv(cfg, B, state="factor", episode_steps=T, seed=77, stagger=True, window_rows=w)
            for w in (0, 0, 100)]
    for env in envs:
        env.reset()
    phase = envs[0].phase.cpu().numpy()
    m_hist = np.zeros(B, dtype=np.int64)
    rewards = []
    for t in range(steps):
        acts = cell_centre_actions(cfg, t, 0, B, B, alts)
        out = []
        for env in envs:
            r, s = env.step(acts)
            assert int(s.abs().sum()) == 0
            out.append(r.clone())
        assert torch.equal(out[0], out[1])
        assert torch.allclose(out[0], out[2], atol=1e-5, rtol=1e-5)  # window covering the whole grid == exact columns
        rewards.append(host(out[0]))
        ranks = host(envs[0].engine.ranks())
        done = (t + 1 + phase) % T == 0  # env e finishes an episode after (t + 1 + phase_e) steps
        assert np.all(ranks[done] == 0) and np.all(ranks[~done] > 0)
        assert np.array_equal(envs[0].episode, 1 + (t + 1 + phase) // T)  # 1: the initial reset of every slot
        prev = host(envs[0].prev)
        assert np.allclose(prev[done], [2.0, 2.0, 14.0]) and np.allclose(prev[~done], acts[~done])
    for e in range(B):
        assert np.array_equal(host(envs[0].ground_truth(e)), host(envs[1].ground_truth(e)))
    # the staged fields are a pure function of (seed, global env id, episode index): a third party can regenerate one
    env = envs[0]
    last_p = env._phase_ending_at(steps - 1)
    ids = env._reset_ids_host[last_p]
    e0 = int(ids[0])
    white = torch.empty((1, dim * dim), dtype=torch.float32, device="cuda")
    env.engine.normal_rows(white, dim * dim, env.seed, env.GT_STREAM + int(env.episode[e0]) - 1, row_ids=[e0],
                           row_offset=env.env_id_offset)
    gt = env.engine.generate_grf(white)
    assert np.array_equal(host(gt[0]), host(env.ground_truth(e0)).reshape(-1))


def test_philox_normals_deterministic_and_standard():
    from ipp_rl_amd import EngineConfig, IPPEngine

    eng = IPPEngine(EngineConfig(x_dim=10, y_dim=10), capacity=2, state="factor", rank_cap=16)
    a = host(eng.normal(1 << 20, seed=5, subsequence=2))
    b = host(eng.normal(1 << 20, seed=5, subsequence=2))
    c = host(eng.normal(1 << 20, seed=5, subsequence=3))
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1.0) < 5e-3 and abs(((a - a.mean()) ** 4).mean() - 3.0) < 0.05
    assert np.array_equal(host(eng.normal(1000, seed=5, subsequence=2)), a[:1000])  # depends on (seed, subseq, i) only


def test_metrics_kernel_vs_oracle():
    dim = 20
    eng, ocfg = make(dim, state="factor")
    rs = np.random.RandomState(8)
    white = rs.normal(size=(1, dim, dim))
    eng.reset(env_ids=[0], white_noise=white)
    st = orc.env_reset(ocfg, white[0])
    prev = np.array([2.0, 2.0, 14.0])
    for _ in range(10):
        a = np.array([4.0 * rs.randint(0, dim) + 2, 4.0 * rs.randint(0, dim) + 2, float(rs.randint(5, 15))])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        eps = rs.normal(size=9)
        eng.step(a[None], prev[None], env_ids=[0], meas_noise=eps[None])
        orc.env_step(ocfg, st, a, eps[:m])
        prev = a
    gt, est, diag = st.gt, st.mean, np.diag(st.P)
    msk = gt.ravel() >= 0.4
    want = np.array([orc.metric_rmse(gt, est), orc.metric_rmse(gt, est, msk), orc.metric_wrmse(gt, est),
                     orc.metric_mll(gt, est, diag), orc.metric_wmll(gt, est, diag), orc.metric_uncertainty(diag),
                     orc.metric_uncertainty(diag, msk), orc.metric_uncertainty_difference(diag, msk)])
    got = host(eng.metrics(env_ids=[0]))[0]
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-4


def test_vec_env_shuffled_priors_vs_oracle():
    """shuffle_prior_cov (mapping/mappings.py:238-240): every episode of every env draws its own (sigma^2, l) in
    [0.8, 1.2] x nominal; the batched driver passes them per reset and the step uses them (checked env by env)."""
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

    dim, B, T = 20, 6, 4
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    env = VecIPPEnv(cfg, B, state="factor", episode_steps=T, seed=5, stagger=True, shuffle_prior_cov=True, window_rows=12)
    env.reset()
    ocfg = orc.OracleConfig(x_dim=dim, y_dim=dim, resolution=cfg.resolution, coeff_a=cfg.coeff_a, coeff_b=cfg.coeff_b)
    for t in range(7):
        acts = cell_centre_actions(cfg, t, 0, B, B, [6.0, 9.0, 14.0])
        _, status = env.step(acts)
        assert int(status.abs().sum()) == 0
    # after 7 steps every env has been reset at least once with a shuffled prior: diag of a just-reset env == its sigma^2
    scales = set()
    for e in range(B):
        steps_in_episode = (7 + int(env.phase[e])) % T
        d = host(env.diag(e))
        if steps_in_episode == 0:
            sv = float(d.max())
            assert abs(float(d.min()) - sv) < 1e-6 and 0.8 * cfg.signal_variance - 1e-6 <= sv <= 1.2 * cfg.signal_variance + 1e-6
            scales.add(round(sv, 6))
        else:
            assert float(d.max()) <= 1.2 * cfg.signal_variance + 1e-6 and float(d.min()) < float(d.max())
    assert len(scales) >= 1
    # one env step by step against the oracle with that env's drawn prior
    e = 0
    env.reset()
    sv, ls = env._prior_scale(np.array([e]), env.episode[e] - 1)[0]  # the draw of the episode that reset() just started
    P = orc.matern_prior(ocfg, sv, ls)
    prev = np.array([2.0, 2.0, 14.0])
    info = {"mean": 0.5 * np.ones((dim, dim)), "value_threshold": 0.4, "interval_factor": 0.0}
    acts = cell_centre_actions(cfg, 50, 0, B, B, [6.0, 9.0, 14.0])
    reward, _ = env.step(acts, auto_reset=False)
    want = orc.predict_step(ocfg, P, prev, acts[e], {"max_v": 2.0, "max_a": 2.0}, info)[0]
    assert abs(float(reward[e]) - want) < TOL


def test_fullsize_windowed_fused_path_tracks_the_exact_mode():
    """BASELINE configs[1] at full size (4096 envs, 50x50, staggered 40-step episodes with their resets): the bench's
    default path (fixed prior -> window 10, fused kernel, ground truths staged on the side stream) against the exact
    factor mode (full columns, k_prepare + k_gain) step by step: rewards within 1e-5, and mean / diag of sampled envs at
    the end.  Same seeds, so both see the same ground truths and measurement noise."""
    import torch
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

    cfg = EngineConfig(x_dim=50, y_dim=50)
    B, T, steps = 4096, 40, 160  # (long enough to have caught the 1-in-10^5 race of the scalar read-back)
    fast = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=77)
    exact = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=0, seed=77)
    assert fast.engine.info.window_rows == 10 and exact.engine.info.window_rows == 0
    fast.reset()
    exact.reset()
    alts = [float(a) for a in range(5, 15)]
    worst = 0.0
    for t in range(steps):
        acts = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts)).cuda()
        r0, s0 = fast.step(acts)
        r1, s1 = exact.step(acts)
        assert int(s0.abs().sum()) == 0 and int(s1.abs().sum()) == 0
        worst = max(worst, float((r0.double() - r1.double()).abs().max()))
        assert torch.equal(fast.engine.ranks(), exact.engine.ranks()), t
    print(f"[4096 envs, {steps} steps] worst |reward(window 10, fused) - reward(exact)| = {worst:.2e}")
    assert worst < 1e-5
    for e in (0, 1, 39, 40, 2047, 4095):
        assert float((fast.mean(e).double() - exact.mean(e).double()).abs().max()) < 1e-5
        assert float((fast.diag(e).double() - exact.diag(e).double()).abs().max()) < 1e-5
        assert torch.equal(fast.ground_truth(e), exact.ground_truth(e))


def test_observe_on_a_patch_engine_after_committed_steps_leaves_the_state_alone():
    """ipp_observe (Sensor.take_measurement, sensors/cameras.py:108-116) on a patch-layout engine whose slots hold columns: the
    observation-only prologue must not gather from the covariance slots (their band-tile addresses do not exist on this layout --
    ADVICE r03: reads past the slot / the arena); same z as a band-tile engine in the same state, state untouched, last env slot
    of a tight arena included."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine

    cfg = EngineConfig(x_dim=50, y_dim=50)
    B = 8
    rs = np.random.RandomState(21)
    gt = torch.as_tensor(rs.uniform(size=(B, cfg.n_cells)), dtype=torch.float32, device="cuda")
    patch = IPPEngine(cfg, capacity=B, state="factor", rank_cap=64, window_rows=-1, fixed_prior=True, max_batch=B)
    band = IPPEngine(cfg, capacity=B, state="factor", rank_cap=64, window_rows=10, tile_threads=256, fixed_prior=True, max_batch=B)
    assert patch.info.patch_layout == 1 and band.info.patch_layout == 0
    prev = np.tile([2.0, 2.0, 14.0], (B, 1))
    for eng in (patch, band):
        eng.reset(gt=gt)
    for t in range(6):
        a = np.stack([4.0 * rs.randint(20, 30, B) + 2.0, 4.0 * rs.randint(20, 30, B) + 2.0, rs.randint(5, 15, B).astype(float)], axis=1)
        eps = rs.normal(size=(B, 9))
        for eng in (patch, band):
            eng.step(a, prev, meas_noise=eps)
        prev = a
    ranks = patch.ranks().clone()
    assert int(ranks.min()) > 0
    before = [(patch.read_mean(e).clone(), patch.read_diag(e).clone()) for e in range(B)]
    cov7 = patch.read_cov(B - 1).clone()
    a = np.stack([4.0 * rs.randint(20, 30, B) + 2.0, 4.0 * rs.randint(20, 30, B) + 2.0, rs.randint(5, 15, B).astype(float)], axis=1)
    eps = rs.normal(size=(B, 9))
    zp, mp, sp = patch.observe(a, meas_noise=eps)
    zb, mb, sb = band.observe(a, meas_noise=eps)
    torch.cuda.synchronize()
    assert torch.equal(zp, zb) and torch.equal(mp, mb) and torch.equal(sp, sb)
    assert torch.equal(patch.ranks(), ranks)
    for e in range(B):
        assert torch.equal(patch.read_mean(e), before[e][0]) and torch.equal(patch.read_diag(e), before[e][1])
    assert torch.equal(patch.read_cov(B - 1), cov7)
