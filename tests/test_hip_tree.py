"""
GPU tests of ipp_tree_step (csrc/k_tree.h): covariance-only predict steps at tree nodes whose
states are the root env's factor columns plus path-local column blocks.  Oracle: the dense restatement of
simulate_prediction_step chained along every path (planning/common/optimization.py:14-30; the tree searches keep
`next_state` as the child's state, planning/mcts_zero/mcts.py:166-265).
"""
import numpy as np
import pytest

from oracle import ipp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
UAV = {"max_v": 2.0, "max_a": 2.0}
D = 6


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def pad(path):
    return list(path) + [-1] * (D - len(path))


@pytest.mark.parametrize("split", ["0", "1"])  # fused k_tree_step / k_tree_prepare + k_tree_gain (default: by launch size)
@pytest.mark.parametrize("dim,window_rows", [(20, 12), (50, 12), (20, 1000)])
def test_tree_steps_vs_chained_oracle_predictions(dim, window_rows, split, monkeypatch):
    from ipp_rl_amd import EngineConfig, IPPEngine

    monkeypatch.setenv("IPP_TREE_SPLIT", split)  # read by ipp_engine_create

    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    ocfg = orc.OracleConfig(x_dim=dim, y_dim=dim, resolution=cfg.resolution, coeff_a=cfg.coeff_a, coeff_b=cfg.coeff_b)
    eng = IPPEngine(cfg, capacity=3, state="factor", rank_cap=128, window_rows=window_rows, node_capacity=32, max_batch=64,
                    score_scratch=True)
    rs = np.random.RandomState(dim)
    white = rs.normal(size=(dim, dim))
    eng.reset(env_ids=[2], white_noise=white[None])
    st = orc.env_reset(ocfg, white)
    prev = np.array([2.0, 2.0, 14.0])
    centre = np.array([dim // 2, dim // 2])

    def random_action():
        c = np.clip(centre + rs.randint(-3, 4, size=2), 0, dim - 1)
        return np.array([4.0 * c[0] + 2.0, 4.0 * c[1] + 2.0, float(rs.choice([6.0, 8.0, 12.0, 14.0]))])

    for _ in range(4):  # the root: a few executed steps (mean moves, so the adaptive mask is not trivial)
        a = random_action()
        eps = rs.normal(size=9)
        eng.step(a[None], prev[None], env_ids=[2], meas_noise=eps[None])
        m = orc.num_measurements(orc.project_fov(ocfg, a), orc.resolution_factor(a))
        orc.env_step(ocfg, st, a, eps[:m])
        prev = a
    root_rank, root_diag, root_cov = eng.rank(2), host(eng.read_diag(2)), host(eng.read_cov(2))
    info = {"mean": st.mean, "value_threshold": 0.4, "interval_factor": 0.0}

    # tree: node id -> (parent id or None, action); states of the oracle per node
    P_of = {None: st.P}
    prev_of = {None: prev}
    path_of = {None: []}
    plan = [(0, None), (1, None), (2, 0), (3, 0), (4, 2), (5, 4), (6, 1), (7, 5), (8, 7)]  # node 8 sits at depth 6
    depth_of = {None: 0}
    for nid, par in plan:
        depth_of[nid] = depth_of[par] + 1
    for level in range(1, 7):  # nodes of one depth are expanded in ONE call (they only depend on shallower nodes)
        batch = [(nid, par) for nid, par in plan if depth_of[nid] == level]
        if not batch:
            continue
        acts = np.array([random_action() for _ in batch])
        prevs = np.array([prev_of[par] for _, par in batch])
        reward, status = eng.tree_step([2] * len(batch), [pad(path_of[par]) for _, par in batch], acts, prevs,
                                       new_ids=[nid for nid, _ in batch])
        assert int(status.abs().sum()) == 0
        for k, (nid, par) in enumerate(batch):
            want, P_new, _, _ = orc.predict_step(ocfg, P_of[par], prevs[k], acts[k], UAV, info)
            assert abs(float(reward[k]) - want) < TOL, (level, nid, float(reward[k]), want)
            P_of[nid], prev_of[nid], path_of[nid] = P_new, acts[k], path_of[par] + [nid]
    for nid in (0, 3, 5, 8):
        assert np.max(np.abs(host(eng.tree_diag(nid)) - np.diag(P_of[nid]))) < TOL

    # predict-only queries at several nodes in one call, nothing recorded
    nodes = [None, 1, 3, 5, 7]
    acts = np.array([random_action() for _ in nodes])
    prevs = np.array([prev_of[nid] for nid in nodes])
    reward, status = eng.tree_step([2] * len(nodes), [pad(path_of[nid]) for nid in nodes], acts, prevs)
    for k, nid in enumerate(nodes):
        want = orc.predict_step(ocfg, P_of[nid], prevs[k], acts[k], UAV, info)[0]
        assert abs(float(reward[k]) - want) < TOL
    # all-candidate scoring from a node state (ipp_tree_score_actions) == per-candidate tree steps == oracle
    cand = np.array([random_action() for _ in range(48)])
    r_all, st_all = eng.tree_score_actions(2, path_of[5], cand, prev_of[5])
    r_each, _ = eng.tree_step([2] * len(cand), [pad(path_of[5])] * len(cand), cand, np.tile(prev_of[5], (len(cand), 1)))
    assert int(st_all.abs().sum()) == 0 and float((r_all - r_each).abs().max()) < TOL
    for k in (0, 7, 23, 47):
        assert abs(float(r_all[k]) - orc.predict_step(ocfg, P_of[5], prev_of[5], cand[k], UAV, info)[0]) < TOL
    # the root env slot is untouched
    assert eng.rank(2) == root_rank and np.array_equal(host(eng.read_diag(2)), root_diag)
    assert np.array_equal(host(eng.read_cov(2)), root_cov)
    assert np.max(np.abs(host(eng.tree_diag(8)) - np.diag(P_of[8]))) < TOL  # and so are the recorded nodes


def test_tree_step_needs_nodes_and_fused_configuration():
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd._ffi import IppError

    cfg = EngineConfig(x_dim=20, y_dim=20)
    plain = IPPEngine(cfg, capacity=2, state="factor", rank_cap=64, window_rows=12)
    plain.reset()
    with pytest.raises(IppError, match="node_capacity"):
        plain.tree_step([0], [pad([])], np.array([[10.0, 10.0, 8.0]]), np.array([[2.0, 2.0, 14.0]]))
    exact = IPPEngine(cfg, capacity=2, state="factor", rank_cap=64, window_rows=0, node_capacity=4)
    exact.reset()
    with pytest.raises(IppError, match="window_rows"):
        exact.tree_step([0], [pad([])], np.array([[10.0, 10.0, 8.0]]), np.array([[2.0, 2.0, 14.0]]))


def test_tree_steps_resolution_one_m25():
    """1 m cells (length scale scaled accordingly): the MC = 25 / VEC = 2 instantiation of the tree kernel."""
    from ipp_rl_amd import EngineConfig, IPPEngine

    dim = 24
    cfg = EngineConfig(x_dim=dim, y_dim=dim, resolution=1.0, length_scale=0.9175)
    ocfg = orc.OracleConfig(x_dim=dim, y_dim=dim, resolution=1.0, coeff_a=cfg.coeff_a, coeff_b=cfg.coeff_b, length_scale=0.9175)
    eng = IPPEngine(cfg, capacity=2, state="factor", rank_cap=160, max_measurements=25, window_rows=12, node_capacity=8, max_batch=4)
    eng.reset()
    P = orc.matern_prior(ocfg)
    mean = 0.5 * np.ones((dim, dim))
    info = {"mean": mean, "value_threshold": 0.4, "interval_factor": 0.0}
    rs = np.random.RandomState(1)
    prev, path = np.array([0.5, 0.5, 5.0]), []
    for nid in range(4):
        a = np.array([rs.randint(8, 16) + 0.5, rs.randint(8, 16) + 0.5, [5.0, 3.0, 4.0, 2.0][nid]])
        reward, status = eng.tree_step([0], [pad(path)], a[None], prev[None], new_ids=[nid])
        want, P, _, _ = orc.predict_step(ocfg, P, prev, a, UAV, info)
        assert int(status[0]) == 0 and abs(float(reward[0]) - want) < TOL, (nid, float(reward[0]), want)
        prev, path = a, path + [nid]
    assert np.max(np.abs(host(eng.tree_diag(3)) - np.diag(P))) < TOL


def test_tree_node_pool_bookkeeping():
    """TreeNodePool: ids, parents, padded paths; predict from a node equals the expansion that follows it."""
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.planning.tree import TreeNodePool

    cfg = EngineConfig(x_dim=20, y_dim=20)
    eng = IPPEngine(cfg, capacity=2, state="factor", rank_cap=96, window_rows=12, node_capacity=16, max_batch=8)
    eng.reset()
    pool = TreeNodePool(eng, 16)
    start = np.array([2.0, 2.0, 14.0])
    a0 = np.array([[38.0, 42.0, 8.0], [10.0, 14.0, 14.0]])
    _, _, kids = pool.expand([0, 0], [None, None], a0, np.tile(start, (2, 1)))
    assert list(kids) == [0, 1] and pool.path(1) == [1, -1, -1, -1, -1, -1]
    a1 = np.array([[42.0, 42.0, 12.0]])
    r_pred, _ = pool.predict([0], [0], a1, a0[:1])
    r_exp, _, grandkid = pool.expand([0], [0], a1, a0[:1])
    assert abs(float(r_pred[0]) - float(r_exp[0])) < 1e-7 and pool.path(int(grandkid[0])) == [0, 2, -1, -1, -1, -1]
    assert int(pool.depth[2]) == 2 and int(pool.parent[2]) == 0 and len(pool) == 3
    # the same action from the sibling's state gives a different reward (different state), from the root yet another
    r_sib, _ = pool.predict([0], [1], a1, a0[1:])
    r_root, _ = pool.predict([0], [None], a1, start[None])
    assert abs(float(r_sib[0]) - float(r_exp[0])) > 1e-4 and abs(float(r_root[0]) - float(r_exp[0])) > 1e-4
