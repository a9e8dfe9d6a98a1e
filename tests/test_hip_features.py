"""
GPU tests of the NN input state planes (ipp_state_plane, csrc/k_plane.h) against the golden
planes recorded from the reference's planning/common/features.py and against the oracle on engine states.
"""
import numpy as np
import pytest

from oracle import ipp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
UAV = {"max_v": 2, "max_a": 2}


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def test_feature_planes_drop_in_vs_reference(golden):
    """The mirrored generate_input_feature_planes (device state planes) reproduces the reference's output, the
    in-place masking of the history included (features.py:98-99)."""
    from tests.params import example_params
    from ipp_rl_amd.mapping.grid_maps import GridMap
    from ipp_rl_amd.mapping.mappings import Mapping
    from ipp_rl_amd.planning.common.features import EpisodeHistory, generate_input_feature_planes
    from ipp_rl_amd.sensors.cameras import RGBCamera
    from ipp_rl_amd.sensors.models.sensor_models import AltitudeSensorModel

    g = golden("features")
    params = example_params(10)
    gm = GridMap(params)
    sensor = RGBCamera(params["sensor"]["field_of_view"], AltitudeSensorModel(0.05, 0.2), gm)
    mapping = Mapping(gm, sensor)
    info = {"mean": g["mean"], "value_threshold": 0.4, "interval_factor": 0}
    for key, k, adaptive, costs in (("planes_full_adaptive_costs", 3, True, True), ("planes_two_plain", 2, False, False),
                                    ("planes_one_adaptive", 1, True, False)):
        hist = EpisodeHistory(3)
        for i in range(k):
            hist.push(g["states"][i].copy(), g["positions"][i].copy(), float(g["budgets"][i]))
        got = generate_input_feature_planes(mapping, hist, 8, 14, info if adaptive else None, UAV, use_action_costs_input=costs)
        assert got.shape == g[key].shape
        assert np.max(np.abs(got - g[key])) < TOL, key
        if adaptive:  # the history's arrays were masked in place like the reference does
            msk = orc.adaptive_mask(g["mean"], g["states"][k - 1], 0.4, 0.0)
            assert np.all(hist.states[0][~msk, :] == 0) and np.all(hist.states[0][:, ~msk] == 0)


@pytest.mark.parametrize("state,window_rows", [("factor", 0), ("factor", 12), ("dense", 0)])
def test_state_plane_of_engine_slots_vs_oracle(state, window_rows):
    """Plane of a stepped env slot (factor slots are densified on the device) vs the oracle's plane of the same
    state, with the slot's own mean and with an explicit mean for the mask."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    dim = 20
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    ocfg = orc.OracleConfig(x_dim=dim, y_dim=dim, resolution=cfg.resolution, coeff_a=cfg.coeff_a, coeff_b=cfg.coeff_b)
    eng = IPPEngine(cfg, capacity=2, state=state, rank_cap=96, window_rows=window_rows, score_scratch=True)
    rs = np.random.RandomState(21)
    white = rs.normal(size=(dim, dim))
    eng.reset(env_ids=[1], white_noise=white[None])
    st = orc.env_reset(ocfg, white)
    prev = np.array([2.0, 2.0, 14.0])
    for t in range(6):
        a = np.array([4.0 * rs.randint(0, dim) + 2.0, 4.0 * rs.randint(0, dim) + 2.0, float(rs.randint(5, 15))])
        eps = rs.normal(size=9)
        eng.step(a[None], prev[None], env_ids=[1], meas_noise=eps[None])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        orc.env_step(ocfg, st, a, eps[:m])
        prev = a
    mask = orc.adaptive_mask(st.mean, st.P, 0.4, 0.0)
    assert 0 < mask.sum() < mask.size
    assert np.max(np.abs(host(eng.state_plane(1)) - orc.state_plane(st.P, mask))) < TOL
    assert np.max(np.abs(host(eng.state_plane(1, adaptive=False)) - orc.state_plane(st.P))) < TOL
    other_mean = rs.uniform(0, 1, size=(dim, dim))
    mask2 = orc.adaptive_mask(other_mean, st.P, 0.4, 0.0)
    assert np.max(np.abs(host(eng.state_plane(1, mean_for_mask=other_mean)) - orc.state_plane(st.P, mask2))) < TOL
