"""
INTER_AREA weights derived ON PAPER from the three clauses of OpenCV 4.5.2's `computeResizeAreaTab` (the algorithm behind
`cv2.resize(..., interpolation=INTER_AREA)` at simulations/sensor_manipulations.py:22), for the per-axis (source -> destination)
sizes the rf = 2 observations of the 50x50 config reach: 2->1, 3->2, 5->3, 4->3, 5->2, 6->3, 3->1, and the identity.

    scale = src / dst;  for destination index d:  a = d * scale, b = a + scale, cell = min(scale, src - a)
    s1 = ceil(a), s2 = min(floor(b), src - 1), s1 = min(s1, s2)
    (1) if s1 - a > 1e-3:  weight (s1 - a) / cell  on source s1 - 1
    (2) for s in [s1, s2): weight 1 / cell          on source s
    (3) if b - s2 > 1e-3:  weight min(min(b - s2, 1), cell) / cell  on source s2

e.g. 5 -> 3 (scale 5/3): d = 1: a = 5/3, b = 10/3, cell = 5/3, s1 = 2, s2 = 3: (1) (2 - 5/3) / (5/3) = 0.2 on source 1, (2) 0.6 on
source 2, (3) (10/3 - 3) / (5/3) = 0.2 on source 3.

cv2 is not installed and not vendored (requirements.txt:8 pins opencv-python==4.5.2.54), so the rf = 2 observation stays PARITY
UNPINNED against the library itself; these tables are the strongest check available here: neither the oracle's
`_area_taps` nor the kernels' `area_weight` is compared with itself.  The CPU test pins the oracle, the GPU test pins
`ipp_observe` (and therefore the fused step's observation, which shares `area_weight`) including the transposed `dsize`
(`dsize=(ceil(h/2), ceil(w/2))` is read as (width, height): the output has ceil(w/2) rows and ceil(h/2) columns).
"""
import numpy as np
import pytest

# TABLES[(src, dst)][d][s]: weight of source index s in destination index d, derived by hand from the clauses above
TABLES = {
    (2, 1): [[1 / 2, 1 / 2]],
    (3, 1): [[1 / 3, 1 / 3, 1 / 3]],
    (3, 2): [[2 / 3, 1 / 3, 0], [0, 1 / 3, 2 / 3]],
    (4, 2): [[1 / 2, 1 / 2, 0, 0], [0, 0, 1 / 2, 1 / 2]],
    (4, 3): [[3 / 4, 1 / 4, 0, 0], [0, 1 / 2, 1 / 2, 0], [0, 0, 1 / 4, 3 / 4]],
    (5, 2): [[2 / 5, 2 / 5, 1 / 5, 0, 0], [0, 0, 1 / 5, 2 / 5, 2 / 5]],
    (5, 3): [[3 / 5, 2 / 5, 0, 0, 0], [0, 1 / 5, 3 / 5, 1 / 5, 0], [0, 0, 0, 2 / 5, 3 / 5]],
    (6, 3): [[1 / 2, 1 / 2, 0, 0, 0, 0], [0, 0, 1 / 2, 1 / 2, 0, 0], [0, 0, 0, 0, 1 / 2, 1 / 2]],
    (1, 1): [[1.0]],
    (2, 2): [[1.0, 0], [0, 1.0]],
    (3, 3): [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]],
}


def table(src, dst):
    return np.asarray(TABLES[(src, dst)], dtype=np.float64)


def expected_observation(sub):
    """cv2.resize(sub, dsize=(ceil(h/2), ceil(w/2)), INTER_AREA) from the hand-derived tables: `dsize` is (width, height), so
    the output has ceil(w/2) rows (taken from the h source rows) and ceil(h/2) columns (from the w source columns)."""
    h, w = sub.shape
    out_rows, out_cols = -(-w // 2), -(-h // 2)
    return table(h, out_rows) @ sub @ table(w, out_cols).T


def test_every_table_row_sums_to_one():
    for key, t in TABLES.items():
        assert np.allclose(np.asarray(t).sum(axis=1), 1.0, atol=1e-15), key


@pytest.mark.parametrize("h,w", [(5, 5), (3, 3), (2, 2), (3, 5), (4, 5), (5, 4), (5, 3), (4, 4), (6, 5), (5, 6), (6, 6), (3, 2), (2, 3)])
def test_oracle_area_resize_matches_the_hand_derived_tables(h, w):
    from oracle import ipp_oracle as orc

    rs = np.random.RandomState(100 * h + w)
    sub = rs.uniform(size=(h, w))
    out_rows, out_cols = -(-w // 2), -(-h // 2)
    if out_rows > h or out_cols > w:  # enlargement along one axis: not reachable from cell-centre actions
        with pytest.raises(NotImplementedError):
            orc.downsample(sub, 2)
        return
    got = orc.downsample(sub, 2)
    assert got.shape == (out_rows, out_cols)  # the dsize transposition of sensor_manipulations.py:22
    # (OpenCV keeps the weights as float32: 1e-7 relative)
    assert np.allclose(got, expected_observation(sub), rtol=0, atol=3e-7)


@pytest.mark.gpu
def test_hip_observation_matches_the_hand_derived_tables():
    """rf = 2 observations (altitudes 11 .. 14 m) of the 50x50 example grid at the centre, the borders and the corners: 5x5,
    3x3 footprints and their clipped shapes 3x5, 4x5, 5x4, 5x3, 3x3, 4x4, 2x2, 3x2 ... through ipp_observe with zero noise."""
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine

    cfg = EngineConfig(x_dim=50, y_dim=50)
    eng = IPPEngine(cfg, capacity=1, state="factor", rank_cap=90, window_rows=-1, fixed_prior=True, max_batch=256)
    rs = np.random.RandomState(11)
    gt = rs.uniform(0.05, 0.95, size=(50, 50))
    eng.reset(env_ids=[0], gt=gt[None])
    res = cfg.resolution
    cells = [0, 1, 2, 3, 24, 46, 47, 48, 49]
    acts, shapes = [], []
    for alt, rad in ((14.0, 2), (12.0, 1)):
        for gy in cells:
            for gx in cells:
                acts.append([res * gx + 0.5 * res, res * gy + 0.5 * res, alt])
                yu, yd = max(gy - rad, 0), min(gy + rad, 49)
                xl, xr = max(gx - rad, 0), min(gx + rad, 49)
                shapes.append((yu, yd, xl, xr))
    acts = np.asarray(acts)
    seen = set()
    for lo in range(0, len(acts), 128):
        a = acts[lo:lo + 128]
        n = len(a)
        z, m, shp = eng.observe(a, env_ids=np.zeros(n, dtype=np.int32), meas_noise=np.zeros((n, 9)))
        z, m, shp = z.cpu().numpy().astype(np.float64), m.cpu().numpy(), shp.cpu().numpy()
        for i in range(n):
            yu, yd, xl, xr = shapes[lo + i]
            sub = gt[yu:yd + 1, xl:xr + 1].astype(np.float32).astype(np.float64)  # (the engine stores the field in fp32)
            want = expected_observation(sub)
            assert tuple(shp[i]) == want.shape and m[i] == want.size, (shapes[lo + i], shp[i], m[i])
            got = z[i, :want.size].reshape(want.shape)
            assert np.allclose(got, want, rtol=0, atol=2e-6), (shapes[lo + i], np.abs(got - want).max())
            seen.add(sub.shape)
    assert {(5, 5), (3, 3), (3, 5), (4, 5), (5, 3), (5, 4), (2, 2), (2, 3), (3, 2), (4, 4), (3, 4), (4, 3)} <= seen
