"""Would a search gain from two groups of roots on two hardware queues?  Two engines of 512 roots each with their own DeviceMCTS
(BASELINE configs[4] shapes, stub network, IPP_MCTS_NOSYNC=1: a wave of simulations is queued without read-backs), searched one
after the other on one stream and concurrently from two host threads on two streams.
    python tools/mcts_two.py [in_flight]"""
import os
import sys
import threading
import time

os.environ.setdefault("IPP_MCTS_NOSYNC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_rl_amd import EngineConfig, IPPEngine
from ipp_rl_amd.planning.mcts_zero.device_mcts import DeviceMCTS
from ipp_rl_amd.vec_env import cell_centre_actions

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
grid, roots, sims, root_steps, horizon = 200, 512, 256, 3, 5
cfg = EngineConfig(x_dim=grid, y_dim=grid)
hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
             dirichlet_alpha=1.0, dirichlet_eps=0.25, num_mcts_simulations=sims)
meta = {"budget": 100.0, "initial_budget": 100.0, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
        "altitude_spacing": 6.0, "uav_specifications": {"max_v": 2.0, "max_a": 2.0},
        "scenario_info": {"value_threshold": 0.4, "interval_factor": 0}}


def make(seed):
    eng = IPPEngine(cfg, capacity=roots, state="factor", rank_cap=9 * (root_steps + horizon + 2), window_rows=-1, fixed_prior=True,
                    node_capacity=roots * (sims + W), max_batch=roots * W, device="cuda:0")
    white = torch.empty((roots, cfg.n_cells), dtype=torch.float32, device="cuda")
    eng.normal_rows(white, cfg.n_cells, seed, 1 << 40)
    eng.reset(white_noise=white)
    prev = np.tile([2.0, 2.0, 14.0], (roots, 1))
    noise = torch.empty((root_steps, roots, eng.meas_cap), dtype=torch.float32, device="cuda")
    eng.normal_rows(noise, eng.meas_cap, seed, 2 << 40)
    for t in range(root_steps):
        a = cell_centre_actions(cfg, t, 0, roots, roots, [8.0, 14.0])
        eng.step(a, prev, meas_noise=noise[t])
        prev = a
    m = DeviceMCTS(eng, hyper, meta, None, sims_in_flight=W, tie_break="random", leaf_value=0.3)
    m.get_policy(list(range(roots)), prev, [100.0] * roots)
    return eng, m, prev


A, B = make(9), make(11)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def search(inst, st):
    eng, m, prev = inst
    with torch.cuda.stream(st):
        out = m.get_policy(list(range(roots)), prev, [100.0] * roots)
        st.synchronize()
    return out


for rep in range(3):
    t0 = time.perf_counter()
    search(A, streams[0]); search(B, streams[0])
    t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=search, args=(x, s)) for x, s in ((A, streams[0]), (B, streams[1]))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t_par = time.perf_counter() - t0
    print(f"{W} in flight, 2 x 512 roots x {sims} simulations: one after the other {t_seq * 1e3:.1f} ms, two threads / two streams {t_par * 1e3:.1f} ms", flush=True)
