"""The plain-C oracle (cpu_baseline "port") against the NumPy oracle and the golden vectors (fp64)."""
import numpy as np
import pytest

from oracle import c_oracle, ipp_oracle as orc


def test_c_prior_matches():
    cfg = orc.OracleConfig(x_dim=12, y_dim=12)
    assert np.max(np.abs(c_oracle.matern_prior(cfg) - orc.matern_prior(cfg))) < 1e-14


@pytest.mark.parametrize("name", ["episode_rf1_20_s1", "episode_mixed_20_s4"])
def test_c_episode_vs_golden(golden, name):
    g = golden(name)
    dim = g["gt"].shape[0]
    cfg = orc.OracleConfig(x_dim=dim, y_dim=dim)
    P = c_oracle.matern_prior(cfg)
    mean = 0.5 * np.ones(dim * dim)
    gt = np.ascontiguousarray(g["gt"]).ravel()
    h = orc.grf_kernel(dim, dim, 5.0)
    assert np.max(np.abs(c_oracle.grf_from_kernel(g["white"], h) - g["gt"])) < 1e-12
    prev = np.array([2.0, 2.0, 14.0])
    for t, a in enumerate(g["actions"]):
        m = int(g["m"][t])
        rc, reward, z = c_oracle.step(cfg, P, mean, gt, a, prev, g["eps"][t][:m])
        assert rc == 0
        assert abs(reward - g["reward"][t]) < 1e-12
        assert np.max(np.abs(z[:m] - g["z"][t][:m])) < 1e-14
        assert np.max(np.abs(mean - g["mean"][t].ravel())) < 1e-12
        assert np.max(np.abs(np.diag(P) - g["diag"][t])) < 1e-12
        prev = a
    if "P_final" in g.files:
        assert np.max(np.abs(P - g["P_final"])) < 1e-12


def test_c_predict_modes_vs_golden(golden):
    g = golden("predict_10")
    cfg = orc.OracleConfig(x_dim=10, y_dim=10)
    P = c_oracle.matern_prior(cfg)
    prev = np.array([2.0, 2.0, 14.0])
    for t, a in enumerate(g["actions"]):
        mode = int(g["mode"][t])
        if mode == 2:
            cfg.value_threshold, cfg.interval_factor = 0.9, 2.0
        else:
            cfg.value_threshold, cfg.interval_factor = 0.4, 0.0
        flags = c_oracle.COV_ONLY | (c_oracle.ADAPTIVE if mode != 1 else 0) | (c_oracle.USE_FLIGHT_TIME if mode in (0, 2) else 0)
        mean = np.ascontiguousarray(g["mean_used"][t]).ravel().copy()
        rc, reward, _ = c_oracle.step(cfg, P, mean, np.zeros(100), a, prev, None, flags)
        assert rc == 0 and abs(reward - g["reward"][t]) < 1e-12
        assert np.max(np.abs(np.diag(P) - g["diag"][t])) < 1e-12
        prev = a
    assert np.max(np.abs(P - g["P_final"])) < 1e-12


def test_c_fallback_matches_numpy():
    cfg = orc.OracleConfig(x_dim=6, y_dim=6, coeff_a=-0.05)  # negative R makes S indefinite -> inverse formula
    P = orc.matern_prior(cfg)
    Pc, mean_c = P.copy(), 0.5 * np.ones(36)
    gt = np.linspace(0, 1, 36)
    act, prev = np.array([10.0, 10.0, 8.0]), np.array([2.0, 2.0, 14.0])
    rc, reward, _ = c_oracle.step(cfg, Pc, mean_c, gt, act, prev, np.zeros(9), 0)
    st = orc.EnvState(mean=0.5 * np.ones((6, 6)), P=P.copy(), gt=gt.reshape(6, 6))
    x, Pn, terms = orc.update_grid_map(cfg, st.P, st.mean, act, orc.observe(cfg, st.gt, act, np.zeros(9)))
    if terms.used_fallback:
        assert rc == 1
        assert np.max(np.abs(Pc - Pn)) < 1e-9 and np.max(np.abs(mean_c - x.ravel())) < 1e-9


def test_c_batch_driver_threads():
    cfg = orc.OracleConfig(x_dim=10, y_dim=10)
    B, T = 6, 4
    rs = np.random.RandomState(0)
    acts = np.stack([4.0 * rs.randint(0, 10, (T, B)) + 2, 4.0 * rs.randint(0, 10, (T, B)) + 2,
                     rs.randint(5, 15, (T, B)).astype(float)], axis=-1)
    gts = rs.uniform(size=(B, 100))
    outs = []
    for threads in (1, 3):
        P = np.stack([c_oracle.matern_prior(cfg)] * B)
        mean = 0.5 * np.ones((B, 100))
        outs.append(c_oracle.run_batch(cfg, P, mean, gts, acts, np.array([2.0, 2.0, 14.0]), threads=threads))
    assert np.array_equal(outs[0], outs[1]) and np.all(outs[0] > 0)
