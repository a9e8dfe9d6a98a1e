"""
GPU tests of the classic MCTS planner (ipp-rl_amd/planning/mcts_mission.py::ClassicMCTS; SURVEY 8(f) row 1, VERDICT r02
missing #3): two searches recorded from the imported reference (planning/mcts_mission.py: run_simulations_proxy with
eps-greedy rollouts on an adaptive mission, with generalised cost-benefit rollouts on a non-adaptive one;
tests/golden/gen_golden.py::gen_mcts_mission) are rebuilt from the same seeds: same root children in the same order
(duplicates included), same visit counts, value sums within 1e-4, same best child -- with every covariance step and every
candidate scoring on the device; then many roots in lock step give the single-root results.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
UAV = {"max_v": 2, "max_a": 2}
TOL = 1e-5


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def build(g, name, capacity=1, node_capacity=2048):
    from ipp_rl_amd import EngineConfig, IPPEngine

    dim, horizon = int(g[f"{name}_dim"]), int(g[f"{name}_horizon"])
    steps = len(g[f"{name}_root_actions"])
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    eng = IPPEngine(cfg, capacity=capacity, state="factor", rank_cap=9 * (steps + horizon + 2), window_rows=-1, fixed_prior=True,
                    node_capacity=node_capacity, max_batch=512, score_scratch=True)
    for slot in range(capacity):
        eng.reset(env_ids=[slot], gt=g[f"{name}_gt"][None])
        prev = np.array([2.0, 2.0, 14.0])
        for a, eps in zip(g[f"{name}_root_actions"], g[f"{name}_root_eps"]):
            _, st = eng.step(a[None], prev[None], env_ids=[slot], meas_noise=eps[None])
            assert int(st[0]) == 0
            prev = a
    assert np.max(np.abs(host(eng.read_mean(0)) - g[f"{name}_root_mean"])) < TOL
    assert np.max(np.abs(host(eng.read_diag(0)) - g[f"{name}_root_diag"])) < TOL
    return cfg, eng


def planner(g, name, cfg, eng):
    from ipp_rl_amd.planning.mcts_mission import ClassicMCTS

    amin, amax, aspc = g[f"{name}_alts"]
    return ClassicMCTS(eng, cfg, UAV, float(amin), float(amax), float(aspc), num_simulations=int(g[f"{name}_sims"]),
                       gamma=float(g["hyper_gamma"]), c=float(g["hyper_c"]), episode_horizon=int(g[f"{name}_horizon"]),
                       k=float(g["hyper_k"]), alpha=float(g["hyper_alpha"]), epsilon_expand=float(g["hyper_epsilon_expand"]),
                       epsilon_rollout=float(g["hyper_epsilon_rollout"]), max_greedy_radius=float(g[f"{name}_radius"]),
                       use_gcb_rollout=bool(g[f"{name}_gcb"]), adaptive=bool(g[f"{name}_adaptive"]))


def tree_size(node):
    return 1 + sum(tree_size(c) for c in node.children)


def tree_depth(node):
    return 0 if not node.children else 1 + max(tree_depth(c) for c in node.children)


def check_root(g, name, root):
    assert root.visits == int(g[f"{name}_root_visits"])
    assert len(root.children) == len(g[f"{name}_child_visits"])
    assert np.array_equal(np.array([c.action for c in root.children]), g[f"{name}_child_actions"])
    assert np.array_equal(np.array([c.visits for c in root.children]), g[f"{name}_child_visits"])
    assert np.array_equal(np.array([len(c.children) for c in root.children]), g[f"{name}_child_children"])
    got = np.array([c.value_sum for c in root.children], dtype=np.float64)
    assert np.allclose(got, g[f"{name}_child_value_sums"], rtol=1e-4, atol=1e-5), np.abs(got - g[f"{name}_child_value_sums"]).max()
    assert abs(root.value_sum - float(g[f"{name}_root_value_sum"])) < 1e-4 * max(1.0, abs(float(g[f"{name}_root_value_sum"])))
    assert tree_size(root) == int(g[f"{name}_tree_nodes"]) and tree_depth(root) == int(g[f"{name}_tree_depth"])


@pytest.mark.parametrize("name", ["eps10", "gcb10"])
def test_search_reproduces_the_reference_mcts_mission(golden, name):
    from ipp_rl_amd.planning.mcts_mission import ClassicMCTS

    g = golden("mcts_mission")
    cfg, eng = build(g, name)
    mcts = planner(g, name, cfg, eng)
    s = mcts.new_search(0, g[f"{name}_prev"], float(g[f"{name}_budget"]), worker_id=0, py_seed=int(g[f"{name}_py_seed"]))
    (root,) = mcts.run([s])
    check_root(g, name, root)
    best = ClassicMCTS.select_best_child(root)
    assert np.array_equal(best.action, g[f"{name}_best_action"])
    print(f"[classic MCTS {name}] {tree_size(root)} tree nodes, {s.nodes} device nodes, {mcts.stats}")


def test_many_roots_in_lock_step_equal_the_single_root_search(golden):
    """Four copies of the recorded root searched together (one ipp_tree_step launch per round for all of them): every root
    reproduces the recorded search, and the launches carry several items."""
    g = golden("mcts_mission")
    name, R = "eps10", 4
    cfg, eng = build(g, name, capacity=R, node_capacity=4096)
    mcts = planner(g, name, cfg, eng)
    searches = [mcts.new_search(r, g[f"{name}_prev"], float(g[f"{name}_budget"]), 0, int(g[f"{name}_py_seed"])) for r in range(R)]
    roots = mcts.run(searches)
    for root in roots:
        check_root(g, name, root)
    assert mcts.stats["device_steps"] > 2 * mcts.stats["launches"]


def test_replan_and_a_short_mission_run():
    """replan -> fly -> measure -> update on the device for a few waypoints: budgets shrink, ranks grow, statuses stay clean."""
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.planning.mcts_mission import ClassicMCTS

    cfg = EngineConfig(x_dim=20, y_dim=20)
    eng = IPPEngine(cfg, capacity=1, state="factor", rank_cap=9 * 12, window_rows=-1, fixed_prior=True, node_capacity=2048,
                    max_batch=512, score_scratch=True)
    rs = np.random.RandomState(2)
    eng.reset(env_ids=[0], white_noise=rs.normal(size=(1, 20, 20)))
    mcts = ClassicMCTS(eng, cfg, UAV, 8.0, 14.0, 6.0, num_simulations=16, episode_horizon=3, max_greedy_radius=9.0, adaptive=True,
                       epsilon_expand=0.2, epsilon_rollout=0.5)
    wps, left = mcts.execute(0, budget=20.0, meas_noise_fn=lambda wp: rs.normal(size=9), max_steps=4)
    assert 1 <= len(wps) <= 4 and left < 20.0
    assert int(eng.rank(0)) > 0 and np.isfinite(host(eng.read_mean(0))).all()
