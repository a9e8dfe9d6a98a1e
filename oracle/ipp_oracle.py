"""
CPU oracle for the IPP environment-step hot path -- TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy restatement of the algorithm the reference
(dmar-bonn/ipp-rl) runs per environment step.  It is the *checker* for the HIP
engine: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``ipp-rl_amd/`` imports it.

Parity status: PINNED for everything except the rf=2 ``INTER_AREA`` downsample
(``area_resize``): that arithmetic lives in opencv-python==4.5.2.54
(reference ``requirements.txt:8``), which is neither vendored in the reference
nor installed in this image, so ``area_resize`` restates OpenCV's published
``computeResizeAreaTab`` algorithm and is "parity unpinned".  Every other
function below was checked against outputs of the imported reference
(``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).

All ``file:line`` citations are relative to the reference repository root.
Conventions: grid is H x W (H = y_dim rows, W = x_dim cols), N = H*W, flat cell
index = W*row + col (C order), position = [x (col axis), y (row axis), altitude].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SQRT3 = math.sqrt(3.0)


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """The hot-path parameters of config/example.yaml."""

    x_dim: int = 50
    y_dim: int = 50
    resolution: float = 4.0
    angle_x: float = 60.0
    angle_y: float = 60.0
    coeff_a: float = 0.05
    coeff_b: float = 0.2
    signal_variance: float = 1.82
    length_scale: float = 3.67
    cluster_radius: float = 5.0
    max_v: float = 2.0
    max_a: float = 2.0
    value_threshold: float = 0.4
    interval_factor: float = 0.0

    @property
    def n_cells(self) -> int:
        return self.x_dim * self.y_dim

    @property
    def uav(self) -> Dict[str, float]:
        return {"max_v": self.max_v, "max_a": self.max_a}

    @classmethod
    def from_params(cls, params: Dict) -> "OracleConfig":
        env, sen, mp = params["environment"], params["sensor"], params["mapping"]
        exp = params.get("experiment", {})
        uav = exp.get("uav", {})
        scen = exp.get("scenario", {})
        return cls(
            x_dim=env["x_dim"],
            y_dim=env["y_dim"],
            resolution=env["resolution"],
            angle_x=sen["field_of_view"]["angle_x"],
            angle_y=sen["field_of_view"]["angle_y"],
            coeff_a=sen["model"]["coeff_a"],
            coeff_b=sen["model"]["coeff_b"],
            signal_variance=mp["signal_variance"],
            length_scale=mp["length_scale"],
            cluster_radius=sen.get("simulation", {}).get("cluster_radius", 5.0),
            max_v=uav.get("max_v", 2.0),
            max_a=uav.get("max_a", 2.0),
            value_threshold=scen.get("value_threshold", 0.4),
            interval_factor=scen.get("interval_factor", 0.0),
        )


# --------------------------------------------------------------------------- sensing (L1)
def fov_range_m(cfg: OracleConfig, altitude: float) -> Tuple[float, float]:
    """Ground-plane extent of the camera frustum. sensors/cameras.py:34-47."""
    ext_x = 2 * altitude * np.tan(0.5 * np.radians(cfg.angle_x))
    ext_y = 2 * altitude * np.tan(0.5 * np.radians(cfg.angle_y))
    return float(ext_x), float(ext_y)


def project_fov(cfg: OracleConfig, pos: Sequence[float]) -> Tuple[int, int, int, int]:
    """Clipped cell rectangle (xl, xr, yu, yd) seen from ``pos``. sensors/cameras.py:49-75."""
    ext_x, ext_y = fov_range_m(cfg, pos[2])
    cells_x = np.floor(ext_x / cfg.resolution)
    cells_y = np.floor(ext_y / cfg.resolution)
    gx = np.floor(pos[0] / cfg.resolution)
    gy = np.floor(pos[1] / cfg.resolution)
    rad_x = np.floor(0.5 * cells_x)
    rad_y = np.floor(0.5 * cells_y)
    xl = min(max(gx - rad_x, 0), cfg.x_dim - 1)
    xr = min(max(gx + rad_x, 0), cfg.x_dim - 1)
    yu = min(max(gy - rad_y, 0), cfg.y_dim - 1)
    yd = min(max(gy + rad_y, 0), cfg.y_dim - 1)
    return int(xl), int(xr), int(yu), int(yd)


def resolution_factor(pos: Sequence[float]) -> int:
    """2 strictly above 10 m, else 1. sensors/cameras.py:122-125."""
    return 2 if pos[2] > 10.0 else 1


def noise_variance(cfg: OracleConfig, pos: Sequence[float]) -> float:
    """a * (1 - exp(-b * altitude)). sensors/models/sensor_models.py:27-30."""
    return float(cfg.coeff_a * (1 - np.exp(-cfg.coeff_b * pos[2])))


def num_measurements(fov: Tuple[int, int, int, int], rf: int) -> int:
    """ceil(w/rf) * ceil(h/rf). mapping/mappings.py:125-126."""
    xl, xr, yu, yd = fov
    return int(np.ceil((xr - xl + 1) / rf) * np.ceil((yd - yu + 1) / rf))


def measurement_noise_scalar(cfg: OracleConfig, pos: Sequence[float], rf: int) -> float:
    """Diagonal entry of R = rf^3 * noise * I. sensors/models/sensor_models.py:32-36."""
    return rf ** 3 * noise_variance(cfg, pos)


def measurement_blocks(cfg: OracleConfig, fov: Tuple[int, int, int, int], rf: int):
    """
    Sparse rows of the measurement model H. sensors/models/sensor_models.py:38-81.

    Returns ``(cells, H_F)``: ``cells`` = flat indices of the f footprint cells in
    row-major footprint order, ``H_F`` = (m, f) weights.  Row i covers the rf x rf
    block (by = i // nx, bx = i % nx) clipped to the footprint; every covered cell
    carries 1/rf^2 when the block is complete, else 1/rf (also for 1-cell blocks).
    """
    xl, xr, yu, yd = fov
    w, h = xr - xl + 1, yd - yu + 1
    nx = (xr - xl) // rf + 1
    m = num_measurements(fov, rf)
    cells = np.array([cfg.x_dim * (yu + ly) + (xl + lx) for ly in range(h) for lx in range(w)], dtype=np.int64)
    H_F = np.zeros((m, w * h))
    for i in range(m):
        by, bx = divmod(i, nx)
        x1 = min(bx * rf + rf, w)
        y1 = min(by * rf + rf, h)
        x0 = min(bx * rf, x1)
        y0 = min(by * rf, y1)
        count = (x1 - x0) * (y1 - y0)
        weight = 1 / rf ** 2 if count >= rf ** 2 else 1 / rf
        for ly in range(y0, y1):
            for lx in range(x0, x1):
                H_F[i, ly * w + lx] = weight
    return cells, H_F


def dense_measurement_matrix(cfg: OracleConfig, fov, rf: int) -> np.ndarray:
    """The full m x N matrix the reference materialises (small grids / tests only)."""
    cells, H_F = measurement_blocks(cfg, fov, rf)
    H = np.zeros((H_F.shape[0], cfg.n_cells))
    H[:, cells] = H_F
    return H


# --------------------------------------------------------------------------- prior (a10)
def cell_centres(cfg: OracleConfig) -> np.ndarray:
    """(N, 2) centres in metres, ordered like product(range(rows), range(cols)). mapping/mappings.py:248-256."""
    rr, cc = np.meshgrid(np.arange(cfg.y_dim), np.arange(cfg.x_dim), indexing="ij")
    return np.stack([rr.ravel(), cc.ravel()], axis=1).astype(np.float64) * cfg.resolution + 0.5 * cfg.resolution


def matern32(dist: np.ndarray, signal_variance: float, length_scale: float) -> np.ndarray:
    """sigma^2 (1 + sqrt3 d/l) exp(-sqrt3 d/l): what the unfitted GPR of mappings.py:242-258 returns."""
    t = SQRT3 * dist / length_scale
    return signal_variance * (1.0 + t) * np.exp(-t)


def matern_prior(cfg: OracleConfig, signal_variance: Optional[float] = None, length_scale: Optional[float] = None):
    """Dense N x N prior covariance of the GP branch of init_priors. mapping/mappings.py:235-261."""
    sv = cfg.signal_variance if signal_variance is None else signal_variance
    ls = cfg.length_scale if length_scale is None else length_scale
    X = cell_centres(cfg)
    d = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    return matern32(d, sv, ls)


def matern_prior_columns(cfg: OracleConfig, cells: np.ndarray, signal_variance=None, length_scale=None) -> np.ndarray:
    """P0[:, cells] without forming P0 (factor form)."""
    sv = cfg.signal_variance if signal_variance is None else signal_variance
    ls = cfg.length_scale if length_scale is None else length_scale
    X = cell_centres(cfg)
    d = np.sqrt(((X[:, None, :] - X[None, cells, :]) ** 2).sum(-1))
    return matern32(d, sv, ls)


def shuffled_prior_scale(cfg: OracleConfig, rng=np.random) -> Tuple[float, float]:
    """The two uniform draws of shuffle_prior_cov, in reference order. mapping/mappings.py:238-240."""
    sv = rng.uniform(low=0.8 * cfg.signal_variance, high=1.2 * cfg.signal_variance)
    ls = rng.uniform(low=0.8 * cfg.length_scale, high=1.2 * cfg.length_scale)
    return float(sv), float(ls)


def random_prior(n_cells: int, prior_cov_mean: float, prior_cov_std: float, rng=np.random) -> np.ndarray:
    """Non-GP branch: A ~ N(mu, sd), P0 = A A^T / ||A||_F. mapping/mappings.py:219-233."""
    A = rng.normal(prior_cov_mean, prior_cov_std, (n_cells, n_cells))
    return (1 / np.linalg.norm(A, ord="fro")) * (A @ A.T)


# --------------------------------------------------------------------------- Kalman update (a9)
@dataclass
class KalmanTerms:
    """Intermediate quantities of one update (exposed for parity tests)."""

    cells: np.ndarray
    H_F: np.ndarray
    S: np.ndarray
    L_upper: Optional[np.ndarray]
    L_inv: Optional[np.ndarray]
    Wc: Optional[np.ndarray]
    used_fallback: bool = False
    S_inv: Optional[np.ndarray] = None
    PHt: Optional[np.ndarray] = None


def kalman_update(
    P: np.ndarray,
    cells: np.ndarray,
    H_F: np.ndarray,
    r_scalar: float,
    mean_flat: Optional[np.ndarray] = None,
    z: Optional[np.ndarray] = None,
    cov_only: bool = False,
):
    """
    Rank-m downdate of the dense covariance + optional mean update. mapping/mappings.py:156-215.

    ``S = H P H^T + R`` is symmetrised, factored as ``S = L^T L`` (L upper), and
    ``Wc = P H^T L^-1``, ``P' = P - Wc Wc^T``, ``x' = x + Wc L^-T (z - H x)``.
    A failed Cholesky falls back to ``P' = P - P H^T S^-1 H P`` (:200-215).
    Returns ``(x' or None, P', KalmanTerms)``.
    """
    m = H_F.shape[0]
    P_FF = P[np.ix_(cells, cells)]
    S = H_F @ (P_FF @ H_F.T) + r_scalar * np.eye(m)
    S = 0.5 * (S + S.T)
    PHt = P[:, cells] @ H_F.T
    try:
        L_upper = np.linalg.cholesky(S).T
        L_inv = np.linalg.inv(L_upper)
        Wc = PHt @ L_inv
        P_new = P - Wc @ Wc.T
        terms = KalmanTerms(cells, H_F, S, L_upper, L_inv, Wc, PHt=PHt)
        if cov_only:
            return None, P_new, terms
        v = np.asarray(z, dtype=np.float64).ravel() - H_F @ mean_flat[cells]
        x_new = mean_flat + Wc @ (L_inv.T @ v)
        return x_new, P_new, terms
    except np.linalg.LinAlgError:
        S_inv = np.linalg.inv(S)
        P_new = P - PHt @ (S_inv @ PHt.T)
        terms = KalmanTerms(cells, H_F, S, None, None, None, used_fallback=True, S_inv=S_inv, PHt=PHt)
        if cov_only:
            return None, P_new, terms
        v = np.asarray(z, dtype=np.float64).ravel() - H_F @ mean_flat[cells]
        x_new = mean_flat + (PHt @ S_inv) @ v
        return x_new, P_new, terms


def update_grid_map(cfg: OracleConfig, P, mean, pos, z=None, cov_only=False):
    """Footprint -> H, R -> Kalman update. mapping/mappings.py:114-153 (the predict_only return form)."""
    rf = resolution_factor(pos)
    fov = project_fov(cfg, pos)
    cells, H_F = measurement_blocks(cfg, fov, rf)
    r_scalar = measurement_noise_scalar(cfg, pos, rf)
    mean_flat = None if mean is None else np.asarray(mean, dtype=np.float64).ravel()
    x_new, P_new, terms = kalman_update(P, cells, H_F, r_scalar, mean_flat, z, cov_only)
    if x_new is not None:
        x_new = x_new.reshape(cfg.y_dim, cfg.x_dim)
    return x_new, P_new, terms


# --------------------------------------------------------------------------- reward (a11-a14)
def flight_time(action, previous_action, max_v: float, max_a: float) -> float:
    """Trapezoidal velocity profile. planning/common/actions.py:32-41."""
    dist = float(np.linalg.norm(np.asarray(action, dtype=np.float64) - np.asarray(previous_action, dtype=np.float64)))
    d_acc = min(dist * 0.5, max_v ** 2 / (2 * max_a))
    d_const = dist - 2 * d_acc
    return d_const / max_v + 2 * math.sqrt(2 * d_acc / max_a)


def action_cost(action, previous_action, uav: Optional[Dict] = None) -> float:
    """Euclidean distance, or flight time when UAV limits are given. planning/common/actions.py:8-16."""
    if uav is None:
        return float(np.linalg.norm(np.asarray(action, dtype=np.float64) - np.asarray(previous_action, dtype=np.float64)))
    return flight_time(action, previous_action, uav["max_v"], uav["max_a"])


def adaptive_mask(mean, P_or_diag, value_threshold: float, interval_factor: float) -> np.ndarray:
    """mean + k * diag(P) >= thr. planning/common/rewards.py:8-12."""
    diag = np.diag(P_or_diag) if np.ndim(P_or_diag) == 2 else np.asarray(P_or_diag)
    return np.asarray(mean, dtype=np.float64).ravel() + interval_factor * diag >= value_threshold


def reward_from_diags(diag_before, diag_after, action, previous_action, uav=None, mask=None) -> float:
    """(sum_mask diag P - sum_mask diag P') / (cost + 1). planning/common/rewards.py:15-31."""
    a, b = np.asarray(diag_before), np.asarray(diag_after)
    if mask is not None:
        a, b = a[mask], b[mask]
    return float((np.sum(a) - np.sum(b)) / (action_cost(action, previous_action, uav) + 1))


def predict_step(cfg: OracleConfig, P, previous_action, action, uav=None, adaptive_info=None):
    """
    mask -> covariance-only predict -> reward; the env-step API every planner calls.
    planning/common/optimization.py:14-30.  Returns (reward, P', KalmanTerms, mask).
    """
    mask = None
    if adaptive_info is not None:
        mask = adaptive_mask(
            adaptive_info["mean"], P, adaptive_info["value_threshold"], adaptive_info["interval_factor"]
        )
    _, P_new, terms = update_grid_map(cfg, P, None, action, cov_only=True)
    rew = reward_from_diags(np.diag(P), np.diag(P_new), action, previous_action, uav, mask)
    return rew, P_new, terms, mask


# --------------------------------------------------------------------------- observation (a15-a18)
def _area_taps(src: int, dst: int) -> List[Tuple[int, int, float]]:
    """
    Per-axis (src index, dst index, weight) taps of OpenCV's area resampler
    (``computeResizeAreaTab`` in modules/imgproc/src/resize.cpp, opencv 4.5.2): weights
    are computed in double and stored as float32.  PARITY UNPINNED (cv2 not installed).
    """
    scale = src / dst
    taps: List[Tuple[int, int, float]] = []
    for d in range(dst):
        a = d * scale
        b = a + scale
        cell = min(scale, src - a)
        s1 = int(math.ceil(a))
        s2 = min(int(math.floor(b)), src - 1)
        s1 = min(s1, s2)
        if s1 - a > 1e-3:
            taps.append((s1 - 1, d, float(np.float32((s1 - a) / cell))))
        for s in range(s1, s2):
            taps.append((s, d, float(np.float32(1.0 / cell))))
        if b - s2 > 1e-3:
            taps.append((s2, d, float(np.float32(min(min(b - s2, 1.0), cell) / cell))))
    return taps


def area_resize(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """
    ``cv2.resize(src, dsize=(width, height), interpolation=INTER_AREA)`` for shrinking scales.
    simulations/sensor_manipulations.py:22 is the call site.  PARITY UNPINNED, see module header.
    """
    dst_w, dst_h = int(dsize[0]), int(dsize[1])
    src_h, src_w = src.shape
    if dst_w > src_w or dst_h > src_h:
        raise NotImplementedError("INTER_AREA enlargement is not on the hot path (SURVEY 8(a) a17)")
    out = np.zeros((dst_h, dst_w))
    tx = _area_taps(src_w, dst_w)
    ty = _area_taps(src_h, dst_h)
    for sy, dy, wy in ty:
        for sx, dx, wx in tx:
            out[dy, dx] += src[sy, sx] * wx * wy
    return out


def downsample(sub: np.ndarray, rf: int) -> np.ndarray:
    """
    Identity at rf=1; at rf=2 the area resampler is asked for dsize=(ceil(h/rf), ceil(w/rf)),
    which OpenCV reads as (width, height) -> output shape (ceil(w/rf), ceil(h/rf)).
    simulations/sensor_manipulations.py:7-26.
    """
    if rf <= 1:
        return sub
    h, w = sub.shape
    d0, d1 = int(math.ceil(h / rf)), int(math.ceil(w / rf))
    return area_resize(sub, (d0, d1))


def observe(cfg: OracleConfig, gt: np.ndarray, pos, unit_noise: np.ndarray) -> np.ndarray:
    """
    crop -> downsample -> + nv * eps -> clip[0,1].  simulations/simulations.py:26-34,
    simulations/sensor_manipulations.py:44-57 (the noise *variance* is used as the std;
    legacy ``normal(0, s)`` equals ``s * standard_normal`` on the same stream).
    ``unit_noise`` holds standard normals of the downsampled shape (or flat, C order).
    """
    xl, xr, yu, yd = project_fov(cfg, pos)
    sub = gt[yu : yd + 1, xl : xr + 1]
    ds = downsample(sub, resolution_factor(pos))
    nv = noise_variance(cfg, pos)
    eps = np.asarray(unit_noise, dtype=np.float64).reshape(ds.shape)
    return np.clip(ds + nv * eps, 0.0, 1.0)


# --------------------------------------------------------------------------- ground truth (a19)
def fft_index_list(n: int) -> List[int]:
    """[0..n/2] + [-(n/2-1)..-1]; one entry short for odd n. simulations/ground_truths.py:7-11."""
    half = n // 2
    return list(range(0, half + 1)) + [-i for i in reversed(range(1, half))]


def grf_amplitude(n_rows: int, n_cols: int, cluster_radius: float) -> np.ndarray:
    """sqrt(k^-c) with 0 at DC; rows/cols beyond the index list stay 0. simulations/ground_truths.py:17-29."""
    amp = np.zeros((n_rows, n_cols))
    for i, kx in enumerate(fft_index_list(n_rows)):
        for j, ky in enumerate(fft_index_list(n_cols)):
            if kx == 0 and ky == 0:
                continue
            amp[i, j] = np.sqrt(np.sqrt(kx ** 2 + ky ** 2) ** (-cluster_radius))
    return amp


def grf_from_white_noise(white: np.ndarray, cluster_radius: float) -> np.ndarray:
    """Re ifft2(fft2(white) * amp), min-max normalised. simulations/ground_truths.py:24-33."""
    amp = grf_amplitude(white.shape[0], white.shape[1], cluster_radius)
    fld = np.fft.ifft2(np.fft.fft2(white) * amp).real
    return (fld - np.min(fld)) / (np.max(fld) - np.min(fld))


def gaussian_random_field(cfg: OracleConfig, rng=np.random) -> np.ndarray:
    """
    simulations/simulations.py:43-47 passes (y_dim, x_dim) into parameters named (x_dim, y_dim),
    so the array shape is (x_dim, y_dim); one normal(size=shape) draw from the global stream.
    """
    white = rng.normal(size=(cfg.x_dim, cfg.y_dim))
    return grf_from_white_noise(white, cfg.cluster_radius)


def grf_kernel(n_rows: int, n_cols: int, cluster_radius: float) -> np.ndarray:
    """Real circular-convolution kernel h = ifft2(amp): field = white (*) h (amp is real and even)."""
    return np.fft.ifft2(grf_amplitude(n_rows, n_cols, cluster_radius)).real


# --------------------------------------------------------------------------- fused env step
@dataclass
class EnvState:
    """One environment: belief (mean + dense covariance), ground truth, last waypoint."""

    mean: np.ndarray
    P: np.ndarray
    gt: np.ndarray
    prev: np.ndarray = field(default_factory=lambda: np.array([2.0, 2.0, 14.0]))  # planning/missions.py:69


def env_reset(cfg: OracleConfig, white: np.ndarray, prior_scale: Optional[Tuple[float, float]] = None) -> EnvState:
    sv, ls = prior_scale if prior_scale is not None else (cfg.signal_variance, cfg.length_scale)
    mean = 0.5 * np.ones((cfg.y_dim, cfg.x_dim))  # mapping/mappings.py:259-260
    return EnvState(mean=mean, P=matern_prior(cfg, sv, ls), gt=grf_from_white_noise(white, cfg.cluster_radius))


def env_step(
    cfg: OracleConfig,
    st: EnvState,
    action,
    unit_noise: Optional[np.ndarray],
    adaptive: bool = True,
    use_flight_time: bool = True,
    cov_only: bool = False,
    commit: bool = True,
):
    """
    simulate_prediction_step + take_measurement + update_grid_map fused into one pass
    (planning/mcts_zero/episode_generators.py:137-146; the covariance downdate the reference
    computes twice is computed once).  Returns dict(reward, z, mean, P, terms, mask, cost).
    """
    action = np.asarray(action, dtype=np.float64)
    info = None
    if adaptive:
        info = {"mean": st.mean, "value_threshold": cfg.value_threshold, "interval_factor": cfg.interval_factor}
    uav = cfg.uav if use_flight_time else None
    reward, P_pred, terms, mask = predict_step(cfg, st.P, st.prev, action, uav, info)
    z = None
    mean_new = st.mean
    if not cov_only:
        z = observe(cfg, st.gt, action, unit_noise)
        mean_new, P_exec, _ = update_grid_map(cfg, st.P, st.mean, action, z)
        assert np.array_equal(P_exec, P_pred)
    out = dict(reward=reward, z=z, mean=mean_new, P=P_pred, terms=terms, mask=mask,
               cost=action_cost(action, st.prev, uav))
    if commit:
        st.mean, st.P, st.prev = mean_new, P_pred, action.copy()
    return out


# --------------------------------------------------------------------------- factor form
@dataclass
class FactorState:
    """P = P0(sv, ls) - U U^T with U = [Wc_1 ... Wc_t]; diag cached."""

    mean: np.ndarray
    U: np.ndarray  # (N, r)
    diag: np.ndarray
    sv: float
    ls: float


def factor_reset(cfg: OracleConfig, prior_scale=None) -> FactorState:
    sv, ls = prior_scale if prior_scale is not None else (cfg.signal_variance, cfg.length_scale)
    n = cfg.n_cells
    return FactorState(mean=0.5 * np.ones(n), U=np.zeros((n, 0)), diag=np.full(n, sv), sv=sv, ls=ls)


def factor_step(cfg: OracleConfig, fs: FactorState, action, z=None, commit=True):
    """Same update as ``kalman_update`` but on the factor state; returns (Wc, y=L^-T v or None)."""
    rf = resolution_factor(action)
    fov = project_fov(cfg, action)
    cells, H_F = measurement_blocks(cfg, fov, rf)
    r_scalar = measurement_noise_scalar(cfg, action, rf)
    P_cols = matern_prior_columns(cfg, cells, fs.sv, fs.ls) - fs.U @ fs.U[cells, :].T  # P[:, F]
    S = H_F @ P_cols[cells, :] @ H_F.T + r_scalar * np.eye(H_F.shape[0])
    S = 0.5 * (S + S.T)
    L_inv = np.linalg.inv(np.linalg.cholesky(S).T)
    Wc = P_cols @ H_F.T @ L_inv
    y = None
    if z is not None:
        y = L_inv.T @ (np.asarray(z).ravel() - H_F @ fs.mean[cells])
    if commit:
        if y is not None:
            fs.mean = fs.mean + Wc @ y
        fs.diag = fs.diag - (Wc ** 2).sum(1)
        fs.U = np.concatenate([fs.U, Wc], axis=1)
    return Wc, y


def factor_to_dense(cfg: OracleConfig, fs: FactorState) -> np.ndarray:
    return matern_prior(cfg, fs.sv, fs.ls) - fs.U @ fs.U.T


# --------------------------------------------------------------------------- evaluation metrics ("next" row 4)
def metric_rmse(gt, est, mask=None) -> float:
    """planning/evaluation_metrics.py:4-13."""
    if mask is None:
        return float(np.sqrt(np.mean(np.square(gt - est))))
    return float(np.sqrt(np.mean(np.square(gt.ravel()[mask] - est.ravel()[mask]))))


def metric_uncertainty(diag, mask=None) -> float:
    """trace / masked trace. planning/evaluation_metrics.py:16-21."""
    return float(np.sum(diag) if mask is None else np.sum(diag[mask]))


def metric_uncertainty_difference(diag, mask) -> float:
    """planning/evaluation_metrics.py:24-28."""
    vi, vu = diag[mask], diag[~mask]
    return float((np.mean(vu) - np.mean(vi)) / np.mean(vu))


def _wrmse_weights(gt, est):
    rng_gt = np.max(gt) - np.min(gt)
    w = (gt - np.min(est)) / rng_gt
    return w / np.sum(w)


def metric_wrmse(gt, est) -> float:
    """planning/evaluation_metrics.py:31-36 (weights use min(est), as written there)."""
    return float(np.sqrt(np.mean(_wrmse_weights(gt, est) * np.square(gt - est))))


def _log_loss(gt, est, diag):
    Pd = diag.reshape(est.shape)
    return 0.5 * np.log(2 * np.pi * Pd) + np.square(gt - est) / 2 * Pd  # operator precedence as in :44


def metric_mll(gt, est, diag) -> float:
    """planning/evaluation_metrics.py:39-45."""
    return float(np.mean(_log_loss(gt, est, diag)))


def metric_wmll(gt, est, diag) -> float:
    """planning/evaluation_metrics.py:48-58."""
    return float(np.mean(_wrmse_weights(gt, est) * _log_loss(gt, est, diag)))


# --------------------------------------------------------------------------- NN input feature planes ("next" row 3)
def min_max_normalize(x: np.ndarray) -> np.ndarray:
    """planning/common/features.py:74-81."""
    lo, hi = np.min(x), np.max(x)
    if lo == hi:
        return x / hi
    return (x - lo) / (hi - lo)


def state_plane(state: np.ndarray, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """Rows / columns outside the adaptive mask zeroed, then min-max normalised. features.py:91-101 (the reference
    zeroes the caller's array in place; this restatement works on a copy)."""
    st = np.array(state, dtype=np.float64, copy=True)
    if mask is not None:
        st[~mask, :] = 0
        st[:, ~mask] = 0
    return min_max_normalize(st)


def costs_plane(cfg: OracleConfig, current_action, min_altitude: float, uav=None) -> np.ndarray:
    """features.py:60-71: row i = cost from the current position (at min_altitude) to the action of index
    i = x_dim * col + row (planning/common/actions.py:71-93), broadcast along the row, min-max normalised."""
    cur = np.array(current_action, dtype=np.float64)
    cur[-1] = min_altitude
    n = cfg.n_cells
    plane = np.zeros((n, n))
    for row in range(cfg.y_dim):
        for col in range(cfg.x_dim):
            a = np.array([cfg.resolution * col + 0.5 * cfg.resolution, cfg.resolution * row + 0.5 * cfg.resolution, min_altitude])
            plane[int(cfg.x_dim * col + row), :] = action_cost(cur, a, uav)
    return min_max_normalize(plane)


def input_feature_planes(cfg: OracleConfig, states, positions, budgets, max_history_length: int, min_altitude: float,
                         max_altitude: float, adaptive_info: Optional[Dict] = None, uav: Optional[Dict] = None,
                         use_action_costs_input: bool = False) -> np.ndarray:
    """generate_input_feature_planes with altitude planes (features.py:83-151).  states[0] is the newest entry of the
    history.  Per entry: [state, x, y, z, budget] planes of shape N x N; missing entries are zero planes."""
    n = cfg.n_cells
    ones = np.ones((n, n))
    planes = []
    for st, pos, budget in zip(states, positions, budgets):
        mask = None
        if adaptive_info is not None:
            mask = adaptive_mask(adaptive_info["mean"], st, adaptive_info["value_threshold"], adaptive_info["interval_factor"])
        extent = cfg.x_dim * cfg.resolution  # features.py:49-50 divides x and y by x_dim * resolution
        planes.extend([state_plane(st, mask), pos[0] / extent * ones, pos[1] / extent * ones,
                       (pos[2] - min_altitude) / (max_altitude - min_altitude) * ones, budget * ones])
    for _ in range(max_history_length - len(states)):
        planes.extend([np.zeros((n, n))] * 5)
    if use_action_costs_input:
        planes.append(costs_plane(cfg, positions[0], min_altitude, uav))
    return np.array(planes)
