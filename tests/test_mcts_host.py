"""
Host-side logic of the tree-search drivers without a GPU: BatchedMCTS (per-root Python, the one pinned against the
reference's MCTS in tests/test_hip_mcts.py) and VectorMCTS (vectorised over the roots) must build the SAME trees when both
break ties by the lowest action index -- on a stand-in engine whose edge rewards are a deterministic, order-independent
function of the measurements taken (like the real covariance state).  Also the geometry helpers against the reference's
dense formulas.
"""
import numpy as np
import pytest

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.planning.mcts_zero.mcts import BatchedMCTS
from ipp_rl_amd.planning.mcts_zero.vector_mcts import VectorMCTS


class _T:
    def __init__(self, a):
        self.a = a

    def detach(self):
        return self

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class MockEngine:
    TREE_DEPTH = 6

    class _c:
        node_capacity = 10 ** 7

    def __init__(self, dim):
        self.cfg = EngineConfig(x_dim=dim, y_dim=dim)
        self.node_actions = {}
        self.steps = 0

    def set_adaptive(self, *a):
        pass

    def set_uav(self, *a):
        pass

    def tree_step(self, roots, paths, acts, prevs, new_ids=None, **kw):
        n = len(roots)
        r = np.zeros(n, dtype=np.float32)
        self.steps += n
        for i in range(n):
            key = [self.node_actions[int(nid)] for nid in paths[i] if nid >= 0]
            base = sum(np.sin(k * 0.37) + 1.3 for k in key)
            ha = acts[i][0] * 0.011 + acts[i][1] * 0.017 + acts[i][2] * 0.13
            r[i] = np.float32((1.5 + np.sin(ha * 7.0 + float(roots[i]))) / (1.0 + 0.3 * len(key)) /
                              (1.0 + np.linalg.norm(acts[i] - prevs[i])) * (1 + 0.01 * base))
            if new_ids is not None and new_ids[i] >= 0:
                self.node_actions[int(new_ids[i])] = ha
        return _T(r), _T(np.zeros(n, dtype=np.int32))


def stub(reqs):
    return [(None, 0.3 + 0.05 * (len(r["valid_idx"]) % 7)) for r in reqs]


def setup(dim, sims, eps, horizon=4, budget=60.0, adaptive=False):
    hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
                 dirichlet_alpha=1.0, dirichlet_eps=eps, num_mcts_simulations=sims)
    meta = {"budget": budget, "initial_budget": budget, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
            "altitude_spacing": 6.0, "uav_specifications": {"max_v": 2, "max_a": 2},
            "scenario_info": {"value_threshold": 0.4, "interval_factor": 0} if adaptive else None}
    return hyper, meta


@pytest.mark.parametrize("dim,R,sims,W,eps", [(12, 6, 40, 1, 0.25), (12, 6, 40, 4, 0.25), (60, 8, 32, 4, 0.0), (200, 12, 40, 4, 0.0)])
def test_vector_driver_builds_the_same_trees_as_the_per_root_driver(dim, R, sims, W, eps):
    """Small action sets (<= 4096: dense reference arithmetic incl. the Dirichlet draw from the per-root generators) and large
    ones (sparse arithmetic; eps = 0 because the two drivers sample the noise of a large action set differently)."""
    hyper, meta = setup(dim, sims, eps)
    rs = np.random.RandomState(1)
    prev = np.stack([4.0 * rs.randint(0, dim, R) + 2, 4.0 * rs.randint(0, dim, R) + 2, np.full(R, 14.0)], 1)
    a = BatchedMCTS(MockEngine(dim), hyper, meta, stub, sims_in_flight=W, tie_break="first", row_costs=True)
    out_a = a.get_policy(list(range(R)), prev, [60.0] * R, rngs=[np.random.RandomState(50 + r) for r in range(R)])
    b = VectorMCTS(MockEngine(dim), hyper, meta, stub, sims_in_flight=W, tie_break="first")
    out_b = b.get_policy(list(range(R)), prev, [60.0] * R, rngs=[np.random.RandomState(50 + r) for r in range(R)])
    for j in range(R):
        nd, rt = a.last_roots[j], int(b.root_ids[j])
        K = int(b.n_K[rt])
        assert np.array_equal(nd.idx, b.t_idx[rt, :K]) and np.array_equal(nd.Nsa, b.t_Nsa[rt, :K]) and nd.Ns == b.n_Ns[rt]
        assert np.allclose(nd.Qsa, b.t_Qsa[rt, :K], rtol=0, atol=1e-12) and np.allclose(nd.Ps, b.t_Ps[rt, :K], rtol=0, atol=1e-15)
        if dim <= 30:  # dense outputs: policy list and valid mask like the reference
            assert np.allclose(out_a[j][0], out_b[j][0], atol=1e-12) and np.array_equal(out_a[j][1], out_b[j][1])
        else:          # sparse outputs: {action: probability}, valid indices
            dense = np.asarray(out_a[j][0])
            assert set(out_b[j][0]) == set(np.nonzero(dense)[0]) and all(abs(dense[i] - p) < 1e-12 for i, p in out_b[j][0].items())
    for key in ("nodes", "inferences", "device_steps", "revisits", "new_visits"):
        assert a.stats[key] == b.stats[key], key
    assert b.stats["launches"] <= a.stats["launches"]


def test_valid_action_sets_match_the_reference_mask():
    hyper, meta = setup(80, 4, 0.0)
    m = VectorMCTS(MockEngine(80), hyper, meta, stub)
    rs = np.random.RandomState(0)
    pos = np.stack([rs.uniform(0, 320, 200), rs.uniform(0, 320, 200), rs.choice([8.0, 11.0, 14.0], 200)], 1)
    pos[:40, :2] = 4.0 * rs.randint(0, 80, (40, 2)) + 2.0  # cell centres, incl. borders
    pos[0, :2], pos[1, :2] = (2.0, 2.0), (318.0, 318.0)
    budget = rs.uniform(1.0, 40.0, 200)
    idx, K = m.valid_sets(pos, budget)
    for i in range(200):
        want = np.nonzero(m.next_actions_mask(pos[i], budget[i]))[0]  # mcts.py:148-158 over all num_actions actions
        assert K[i] == len(want) and np.array_equal(idx[i, : K[i]], want) and np.all(idx[i, K[i]:] == -1)
        got, _ = m._valid_idx(pos[i], budget[i])
        assert np.array_equal(got, want)


def test_parallel_simulations_keep_the_accounting_consistent():
    hyper, meta = setup(40, 64, 0.25, horizon=5, budget=100.0)
    R = 32
    rs = np.random.RandomState(3)
    prev = np.stack([4.0 * rs.randint(0, 40, R) + 2, 4.0 * rs.randint(0, 40, R) + 2, np.full(R, 14.0)], 1)
    eng = MockEngine(40)
    m = VectorMCTS(eng, hyper, meta, stub, sims_in_flight=4, tie_break="random", seed=5)
    out = m.get_policy(list(range(R)), prev, [100.0] * R)
    for j in range(R):
        rt = int(m.root_ids[j])
        K = int(m.n_K[rt])
        assert m.n_Ns[rt] == m.t_Nsa[rt, :K].sum() == 60 and np.all(m.t_Nsa[rt] >= 0)  # 64 simulations, the first wave of 4 expands the root
        assert out[j] is not None and abs(sum(out[j][0]) - 1.0) < 1e-9
    assert np.all(m.t_Nsa[: m.n_count].sum(axis=1) == m.n_Ns[: m.n_count])          # virtual visits all undone
    assert not np.any(np.isinf(m.t_num[: m.n_count]))                                # every requested edge got its device result
    assert eng.steps == m.stats["device_steps"] and m.stats["launches"] <= 16 * 6    # <= one launch per level and wave


def test_merge_roots_counts_an_action_once_like_the_reference():
    """planning/mcts_mission.py:320-339: both roots' children are put into dicts keyed by the action first -- of several children
    with the same action (expansion does not de-duplicate) the LAST one stands for the action, and every action of root_b is merged
    or appended once."""
    from ipp_rl_amd.planning.mcts_mission import ClassicMCTS, Node

    def node(action, visits, value_sum):
        n = Node(0, -1, [], action=action)
        n.visits, n.value_sum = visits, value_sum
        return n

    root_a, root_b = node([0, 0, 0], 10, 5.0), node([0, 0, 0], 7, 3.0)
    a1, a2 = node([1, 1, 8], 4, 2.0), node([2, 2, 8], 6, 3.0)
    root_a.children = [a1, a2]
    b1, b1_dup, b3, b3_dup = node([1, 1, 8], 1, 0.5), node([1, 1, 8], 2, 1.5), node([3, 3, 8], 3, 0.75), node([3, 3, 8], 1, 0.25)
    root_b.children = [b1, b1_dup, b3, b3_dup]
    out = ClassicMCTS.merge_roots(root_a, root_b)
    assert out is root_a
    assert (a1.visits, a1.value_sum) == (4 + 2, 2.0 + 1.5)          # the last duplicate of root_b, once
    assert (a2.visits, a2.value_sum) == (6, 3.0)
    assert len(root_a.children) == 3 and root_a.children[2] is b3_dup  # appended once
    assert (root_a.visits, root_a.value_sum) == (10 + 2 + 1, 5.0 + 1.5 + 0.25)
