"""
GPU parity tests: the HIP engine (through the C-ABI / ctypes) against
  (1) golden vectors recorded from the imported reference (tests/golden, fp64), and
  (2) the CPU oracle (oracle/ipp_oracle.py) on the same seeded inputs.
Tolerance: 1e-5 absolute per cell (BASELINE.json north_star: "cell-for-cell within 1e-5 fp32").
"""
import numpy as np
import pytest

from oracle import ipp_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


def engine_for(dim, state, capacity=2, **kw):
    from ipp_rl_amd import EngineConfig, IPPEngine

    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    eng = IPPEngine(cfg, capacity=capacity, state=state, rank_cap=kw.pop("rank_cap", 384), **kw)
    eng.debug_capture()  # (S / z of every step for debug_item)
    return eng


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("state", ["dense", "factor"])
@pytest.mark.parametrize("tag,dim", [("10", 10), ("50", 50)])
def test_predict_sequence_vs_golden(golden, state, tag, dim):
    """simulate_prediction_step sequence (planning/common/optimization.py:14-30): reward, S, diag, rows of P'."""
    g = golden(f"predict_{tag}")
    eng = engine_for(dim, state)
    eng.reset(env_ids=[0])
    prev = np.array([2.0, 2.0, 14.0])
    worst = dict(reward=0.0, diag=0.0, S=0.0, rows=0.0)
    for t, a in enumerate(g["actions"]):
        mode = int(g["mode"][t])
        if mode in (0, 3):
            eng.set_adaptive(0.4, 0.0)
        elif mode == 2:
            eng.set_adaptive(0.9, 2.0)
        if mode != 1:
            eng.write_mean(0, g["mean_used"][t])
        reward, status = eng.step(a[None], prev[None], env_ids=[0], cov_only=True, adaptive=(mode != 1),
                                  use_flight_time=mode in (0, 2))
        it = eng.debug_item(0)
        m = int(g["m"][t])
        assert int(status[0]) == 0
        assert it["fov"] == tuple(g["fov"][t]) and it["rf"] == g["rf"][t] and it["m"] == m
        assert abs(it["cost"] - g["cost"][t]) < 1e-12
        worst["S"] = max(worst["S"], np.max(np.abs(it["S"] - g["S"][t][:m, :m])))
        worst["reward"] = max(worst["reward"], abs(float(reward[0]) - g["reward"][t]))
        worst["diag"] = max(worst["diag"], np.max(np.abs(host(eng.read_diag(0)) - g["diag"][t])))
        if t % 4 == 0 or t == len(g["actions"]) - 1:
            P = host(eng.read_cov(0))
            worst["rows"] = max(worst["rows"], np.max(np.abs(P[g["sample_rows"]] - g["rows"][t])))
            if "P_seq" in g.files and t < len(g["P_seq"]):
                worst["rows"] = max(worst["rows"], np.max(np.abs(P - g["P_seq"][t])))
        prev = a
    print(f"[{state} {dim}x{dim}] worst abs errors: {worst}")
    assert worst["S"] < TOL and worst["reward"] < TOL and worst["diag"] < TOL and worst["rows"] < TOL


EPISODES = ["episode_rf1_20_s0", "episode_rf1_20_s2", "episode_mixed_20_s4", "episode_rf1_50_s0", "episode_mixed_50_s1"]


@pytest.mark.parametrize("state", ["dense", "factor"])
@pytest.mark.parametrize("name", EPISODES)
def test_episode_vs_golden(golden, state, name):
    """40 fused env steps (predict + observe + update): reward, z, mean, diag per step; P and metrics at the end."""
    g = golden(name)
    dim = g["gt"].shape[0]
    eng = engine_for(dim, state)
    eng.reset(env_ids=[0], white_noise=g["white"][None])  # device GRF from the recorded white noise
    gt_dev = host(eng.read_gt(0))
    assert np.max(np.abs(gt_dev - g["gt"])) < TOL
    prev = np.array([2.0, 2.0, 14.0])
    worst = dict(reward=0.0, z=0.0, mean=0.0, diag=0.0)
    for t, a in enumerate(g["actions"]):
        m = int(g["m"][t])
        reward, status = eng.step(a[None], prev[None], env_ids=[0], meas_noise=g["eps"][t][None])
        assert int(status[0]) == 0
        it = eng.debug_item(0)
        assert it["m"] == m and it["rf"] == g["rf"][t]
        worst["reward"] = max(worst["reward"], abs(float(reward[0]) - g["reward"][t]))
        worst["z"] = max(worst["z"], np.max(np.abs(it["z"] - g["z"][t][:m])))
        worst["mean"] = max(worst["mean"], np.max(np.abs(host(eng.read_mean(0)) - g["mean"][t])))
        worst["diag"] = max(worst["diag"], np.max(np.abs(host(eng.read_diag(0)) - g["diag"][t])))
        prev = a
    P = host(eng.read_cov(0))
    err_rows = np.max(np.abs(P[g["sample_rows"]] - g["P_final_rows"]))
    if "P_final" in g.files:
        err_rows = max(err_rows, np.max(np.abs(P - g["P_final"])))
    met = host(eng.metrics(env_ids=[0]))[0]
    err_met = np.max(np.abs(met - g["metrics"]) / np.maximum(1.0, np.abs(g["metrics"])))
    print(f"[{state} {name}] worst abs errors: {worst} P={err_rows:.2e} metrics(rel)={err_met:.2e}")
    assert max(worst.values()) < TOL and err_rows < TOL
    assert err_met < 1e-4  # eight scalar reductions over N cells, fp32 inputs
    if state == "factor":
        assert eng.rank(0) == int(np.sum(g["m"]))


@pytest.mark.parametrize("state", ["dense", "factor"])
def test_batch_vs_oracle_random(state):
    """B independent envs with different ground truths / actions / prior scales vs the fp64 oracle."""
    dim, B, T = 16, 12, 10
    cfg = orc.OracleConfig(x_dim=dim, y_dim=dim)
    eng = engine_for(dim, state, capacity=B)
    rs = np.random.RandomState(42)
    white = rs.normal(size=(B, dim, dim))
    scales = np.stack([rs.uniform(0.8 * 1.82, 1.2 * 1.82, B), rs.uniform(0.8 * 3.67, 1.2 * 3.67, B)], axis=1)
    eng.reset(white_noise=white, prior_scale=scales)
    envs = [orc.env_reset(cfg, white[b], tuple(scales[b])) for b in range(B)]
    prev = np.tile(np.array([2.0, 2.0, 14.0]), (B, 1))
    for t in range(T):
        acts = np.stack([4.0 * rs.randint(0, dim, B) + 2.0, 4.0 * rs.randint(0, dim, B) + 2.0,
                         rs.randint(5, 15, B).astype(float)], axis=1)
        eps = rs.normal(size=(B, 9)) if t % 2 == 0 else None  # odd steps exercise the noise-free (NULL) path
        reward, status = eng.step(acts, prev, meas_noise=eps)
        assert int(status.abs().sum()) == 0
        for b in range(B):
            m = orc.num_measurements(orc.project_fov(cfg, acts[b]), orc.resolution_factor(acts[b]))
            out = orc.env_step(cfg, envs[b], acts[b], eps[b, :m] if eps is not None else np.zeros(m))
            assert abs(float(reward[b]) - out["reward"]) < TOL
        prev = acts
    for b in range(B):
        assert np.max(np.abs(host(eng.read_mean(b)) - envs[b].mean)) < TOL
        assert np.max(np.abs(host(eng.read_diag(b)) - np.diag(envs[b].P))) < TOL
    P = host(eng.read_cov(B - 1))
    assert np.max(np.abs(P - envs[B - 1].P)) < TOL


def test_dense_and_factor_agree():
    """factor-form == dense-form on the device itself."""
    dim = 20
    d, f = engine_for(dim, "dense"), engine_for(dim, "factor")
    rs = np.random.RandomState(1)
    white = rs.normal(size=(1, dim, dim))
    d.reset(env_ids=[0], white_noise=white)
    f.reset(env_ids=[0], white_noise=white)
    prev = np.array([[2.0, 2.0, 14.0]])
    for t in range(25):
        a = np.array([[4.0 * rs.randint(0, dim) + 2.0, 4.0 * rs.randint(0, dim) + 2.0, float(rs.randint(5, 15))]])
        eps = rs.normal(size=(1, 9))
        rd, _ = d.step(a, prev, env_ids=[0], meas_noise=eps)
        rf_, _ = f.step(a, prev, env_ids=[0], meas_noise=eps)
        assert abs(float(rd[0]) - float(rf_[0])) < TOL
        prev = a
    assert np.max(np.abs(host(d.read_cov(0)) - host(f.read_cov(0)))) < TOL
    assert np.max(np.abs(host(d.read_mean(0)) - host(f.read_mean(0)))) < TOL
