"""
GPU parity on grids beyond the golden fixtures' 50x50: the oracle's factor form (P = P0 - U U^T, fp64, pinned
against the dense restatement and the golden vectors in tests/test_oracle_golden.py) scales to N = 14 400, so the
large-grid code paths can be checked cell for cell:
  * N > 12 288: no prior table in LDS (direct sqrt/exp in the base term),
  * many tiles per env, windowed column spans, every streaming kernel (fused, workgroup, wave, exact),
  * GRF: DFT path (even n <= 100) and convolution fallback (n > 100, odd n).
"""
import numpy as np
import pytest

from oracle import ipp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
UAV = {"max_v": 2.0, "max_a": 2.0}


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def make(dim, window_rows, tile_threads, capacity, rank_cap=72):
    from ipp_rl_amd import EngineConfig, IPPEngine

    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    eng = IPPEngine(cfg, capacity=capacity, state="factor", rank_cap=rank_cap, window_rows=window_rows,
                    tile_threads=tile_threads, fixed_prior=window_rows < 0)
    ocfg = orc.OracleConfig(x_dim=dim, y_dim=dim, resolution=cfg.resolution, coeff_a=cfg.coeff_a, coeff_b=cfg.coeff_b)
    return eng, cfg, ocfg


@pytest.mark.parametrize("dim,window_rows,tile_threads", [
    (64, 12, 256), (64, 12, 128), (64, 12, 64), (64, 0, 0),
    (120, 12, 256), (120, 12, 128), (120, 12, 64), (120, 0, 0), (120, 1000, 256),
    (200, 12, 256), (200, 0, 0),  # BASELINE configs[4] grid size
    (100, -1, 256), (100, 0, 0),  # BASELINE configs[2] grid size, whole 16-step episodes (bench window / exact mode)
])
def test_factor_step_vs_oracle_factor_form(dim, window_rows, tile_threads):
    """B envs x 6 steps (100x100: a whole 16-step episode of configs[2]) with clustered revisits (so that stored columns
    are streamed on the footprint tiles), all altitude classes, border footprints; reward / mean / diag against the fp64
    factor-form oracle."""
    B, steps = 3, (16 if dim == 100 else 6)
    eng, cfg, ocfg = make(dim, window_rows, tile_threads, B, rank_cap=9 * steps + 18)
    rs = np.random.RandomState(100 + dim)
    gts = rs.uniform(0.0, 1.0, size=(B, dim, dim))
    eng.reset(gt=gts)
    fss = [orc.factor_reset(ocfg) for _ in range(B)]
    prev = np.tile(np.array([2.0, 2.0, 14.0]), (B, 1))
    res = cfg.resolution
    centres = rs.randint(3, dim - 3, size=(B, 2))
    centres[0] = (0, dim - 1)  # env 0 works in a corner: clipped footprints
    worst = 0.0
    for t in range(steps):
        acts = np.empty((B, 3))
        for b in range(B):
            col = np.clip(centres[b, 0] + rs.randint(-2, 3), 0, dim - 1)
            row = np.clip(centres[b, 1] + rs.randint(-2, 3), 0, dim - 1)
            acts[b] = (res * col + 0.5 * res, res * row + 0.5 * res, [5.0, 8.0, 12.0, 14.0, 9.0, 14.0][(t + b) % 6])
        eps = rs.normal(size=(B, 9))
        reward, status = eng.step(acts, prev, meas_noise=eps)
        assert int(status.abs().sum()) == 0
        for b in range(B):
            fs = fss[b]
            m = orc.num_measurements(orc.project_fov(ocfg, acts[b]), orc.resolution_factor(acts[b]))
            z = orc.observe(ocfg, gts[b], acts[b], eps[b, :m])
            mask = orc.adaptive_mask(fs.mean, fs.diag, 0.4, 0.0)
            diag_before = fs.diag.copy()
            orc.factor_step(ocfg, fs, acts[b], z=z)
            want = orc.reward_from_diags(diag_before, fs.diag, acts[b], prev[b], UAV, mask)
            worst = max(worst, abs(float(reward[b]) - want))
            assert abs(float(reward[b]) - want) < TOL, (t, b, float(reward[b]), want)
        prev = acts
    for b in range(B):
        assert np.max(np.abs(host(eng.read_mean(b)).ravel() - fss[b].mean)) < TOL
        assert np.max(np.abs(host(eng.read_diag(b)).ravel() - fss[b].diag)) < TOL
        assert eng.rank(b) == fss[b].U.shape[1]
    print(f"[{dim}x{dim}, window {window_rows}, T={tile_threads}] worst reward error {worst:.2e}")


@pytest.mark.parametrize("n", [10, 36, 50, 64, 74, 98, 100, 120, 150, 200, 254, 256, 258, 15])
def test_grf_sizes_vs_oracle(n):
    """Even n <= 256: half-spectrum DFT kernel (256 threads up to n = 100, 1024 threads above; 2 or 4 columns per thread:
    the list covers every instantiation); n = 258 and odd n (the
    reference's amplitude table loses its last row / column there, ground_truths.py:8-11): circular-convolution
    kernel.  All against numpy's FFT path."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    eng = IPPEngine(EngineConfig(x_dim=n, y_dim=n), capacity=3, state="factor", rank_cap=16)
    rs = np.random.RandomState(n)
    white = rs.normal(size=(3, n, n))
    out = host(eng.generate_grf(white)).reshape(3, n, n)
    for k in range(3):
        ref = orc.grf_from_white_noise(white[k], 5.0)
        assert np.max(np.abs(out[k] - ref)) < TOL, (n, k, np.max(np.abs(out[k] - ref)))
        assert out[k].min() == 0.0 and abs(out[k].max() - 1.0) < 1e-6
    eng.reset(env_ids=[1], white_noise=white[2][None])
    assert np.max(np.abs(host(eng.read_gt(1)) - orc.grf_from_white_noise(white[2], 5.0))) < TOL
