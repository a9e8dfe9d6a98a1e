"""Where the host time of one device search goes (BASELINE configs[4] shapes, stub network): cProfile of DeviceMCTS.get_policy after a
warm-up search, and wall times of the search with the policies as per-root dicts (the reference's format) and as arrays.
    python tools/mcts_readout.py [in_flight] [roots]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_rl_amd import EngineConfig, IPPEngine
from ipp_rl_amd.planning.mcts_zero.device_mcts import DeviceMCTS
from ipp_rl_amd.vec_env import cell_centre_actions

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
grid, sims, root_steps, horizon = 200, 256, 3, 5
cfg = EngineConfig(x_dim=grid, y_dim=grid)
hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
             dirichlet_alpha=1.0, dirichlet_eps=0.25, num_mcts_simulations=sims)
meta = {"budget": 100.0, "initial_budget": 100.0, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
        "altitude_spacing": 6.0, "uav_specifications": {"max_v": 2.0, "max_a": 2.0},
        "scenario_info": {"value_threshold": 0.4, "interval_factor": 0}}
eng = IPPEngine(cfg, capacity=R, state="factor", rank_cap=9 * (root_steps + horizon + 2), window_rows=-1, fixed_prior=True,
                node_capacity=R * (sims + W), max_batch=R * W, device="cuda:0")
white = torch.empty((R, cfg.n_cells), dtype=torch.float32, device="cuda")
eng.normal_rows(white, cfg.n_cells, 1, 1 << 40)
eng.reset(white_noise=white)
prev = np.tile([2.0, 2.0, 14.0], (R, 1))
noise = torch.empty((root_steps, R, eng.meas_cap), dtype=torch.float32, device="cuda")
eng.normal_rows(noise, eng.meas_cap, 1, 2 << 40)
for t in range(root_steps):
    a = cell_centre_actions(cfg, t, 0, R, R, [8.0, 14.0])
    eng.step(a, prev, meas_noise=noise[t])
    prev = a
m = DeviceMCTS(eng, hyper, meta, None, sims_in_flight=W, tie_break=os.environ.get("TIE", "random"), leaf_value=0.3,
               **({"groups": int(os.environ["MCTS_GROUPS"])} if os.environ.get("MCTS_GROUPS") else {}))
roots, budgets = list(range(R)), [100.0] * R
for _ in range(2):
    m.get_policy(roots, prev, budgets)
torch.cuda.synchronize()
print("groups", m.groups, "queue_ahead", m.queue_ahead, "patch", int(eng.info.patch_layout), "max_batch", eng.max_batch, "R*W", R * m.sims_in_flight,
      "actions", m.num_actions, m.DENSE_ACTIONS, flush=True)
print("groups used:", len(m._subs_used) if getattr(m, "_subs_used", None) else 1, flush=True)
for kw in ({}, {"as_arrays": True}):
    if kw and "as_arrays" not in DeviceMCTS.get_policy.__code__.co_varnames:
        break
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.get_policy(roots, prev, budgets, **kw)
        ts.append(time.perf_counter() - t0)
    print(f"get_policy{kw}: " + " ".join(f"{t * 1e3:.2f}" for t in ts) + " ms")
pr = cProfile.Profile()
pr.enable()
m.get_policy(roots, prev, budgets)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
if os.environ.get("IPP_TIMELINE_FILE"):  # (-DIPP_TIMELINE=1 build: marks of the last tree-step launch -> tools/timeline.py)
    eng.streamed_bytes()
