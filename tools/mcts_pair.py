"""Two groups of roots on two hardware queues, ONE host thread: two engines of R/2 roots each with their own DeviceMCTS (BASELINE
configs[4] shapes, stub network), their waves of simulations issued alternately on two streams of different hardware queues --
against the same two searches one after the other and against one search of R roots.  (tools/mcts_two.py used two host threads and
unclassed streams: the GIL and possibly one shared queue.)       python tools/mcts_pair.py [in_flight] [roots]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_rl_amd import EngineConfig, IPPEngine, _ffi
from ipp_rl_amd.planning.mcts_zero.device_mcts import DeviceMCTS
from ipp_rl_amd.vec_env import cell_centre_actions

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R_all = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NG = int(sys.argv[3]) if len(sys.argv) > 3 else 2  # groups of roots (one engine and one stream each)
grid, sims, root_steps, horizon = 200, 256, 3, 5
cfg = EngineConfig(x_dim=grid, y_dim=grid)
hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
             dirichlet_alpha=1.0, dirichlet_eps=0.25, num_mcts_simulations=sims)
meta = {"budget": 100.0, "initial_budget": 100.0, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
        "altitude_spacing": 6.0, "uav_specifications": {"max_v": 2.0, "max_a": 2.0},
        "scenario_info": {"value_threshold": 0.4, "interval_factor": 0}}


def make(roots, seed):
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
    m = DeviceMCTS(eng, hyper, meta, None, sims_in_flight=W, tie_break="random", leaf_value=0.3, groups=1)  # (this tool does the grouping itself)
    m.get_policy(list(range(roots)), prev, [100.0] * roots)  # (allocates the tables)
    return eng, m, prev


class Search:
    """The no-read-back wave loop of DeviceMCTS.get_policy, one wave per call (so that two searches can alternate)."""

    def __init__(self, inst, stream):
        self.eng, self.m, prev = inst
        m, eng = self.m, self.eng
        self.R = R = eng.capacity
        dev = eng.device
        self.D = m.horizon + 1
        self.tab, self.b = m._tab, m._buf
        self.prev0 = torch.as_tensor(np.asarray(prev, dtype=np.float64).reshape(R, 3), device=dev)
        self.budget0 = torch.full((R,), 100.0, dtype=torch.float64, device=dev)
        self.root_env = torch.arange(R, dtype=torch.int32, device=dev)
        self.stream = stream
        self.cs = C.c_void_p(stream.cuda_stream)

    def begin(self):
        b, R, npr = self.b, self.R, self.m.nodes_per_root
        with torch.cuda.stream(self.stream):
            root_nodes = torch.arange(R, device=self.eng.device, dtype=torch.int64) * npr
            b["n_flags"].zero_(); b["n_flags"][root_nodes] = 2
            b["n_value"].zero_(); b["n_devpath"].fill_(-1)
            b["n_hash"][root_nodes] = (torch.arange(R, device=self.eng.device, dtype=torch.int64) + 1) * (-7046029254386353131)
            b["root_count"].fill_(1); b["dev_count"].zero_(); b["h_keys"].zero_(); b["err"].zero_(); b["counts"].zero_()
        self.sim = 0

    def wave(self):
        m, lib, tp = self.m, self.eng._lib, C.byref(self.tab)
        w = min(W, sims - self.sim)
        flags = _ffi.IPP_ADAPTIVE | _ffi.IPP_USE_FLIGHT_TIME
        with torch.cuda.stream(self.stream):
            _ffi.check(lib.ipp_mcts_select(tp, self.root_env.data_ptr(), self.prev0.data_ptr(), self.budget0.data_ptr(), 0, int(self.sim), int(w),
                                           C.c_uint64(m.seed & (2 ** 64 - 1)), self.cs))
            _ffi.check(lib.ipp_mcts_steps(self.eng._h, tp, 0, -1, flags, self.cs))
            m._expand(lib, tp, self.b, self.R, W, self.root_env, self.cs)
            _ffi.check(lib.ipp_mcts_backup(tp, int(w), self.cs))
        self.sim += w
        return self.sim < sims


def pick_two(eng):
    """two streams on different hardware queues (and off the caller's), by ipp_probe_stream_pair"""
    main = torch.cuda.current_stream()
    thr = 0.75 * min(eng.probe_stream_pair(main, main, 12) for _ in range(2))
    got = []
    for _ in range(12):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda").add_(1)
        if eng.probe_stream_pair(st, main, 12) > thr:
            continue
        if all(eng.probe_stream_pair(st, g, 12) <= thr for g in got):
            got.append(st)
        if len(got) == NG:
            break
    while len(got) < NG:  # (fewer hardware queues than groups: the rest share)
        got.append(got[len(got) % max(1, len(got))] if got else main)
    return got


whole = make(R_all, 9)
groups = [make(R_all // NG, 9 + 2 * g) for g in range(NG)]
torch.cuda.synchronize()
streams = pick_two(groups[0][0])
print(f"{len(set(id(x) for x in streams))} distinct streams for {NG} groups", flush=True)
for rep in range(3):
    s0 = Search(whole, streams[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s0.begin()
    while s0.wave():
        pass
    torch.cuda.synchronize(); t_whole = time.perf_counter() - t0
    ss = [Search(g, streams[0]) for g in groups]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in ss:
        s.begin()
        while s.wave():
            pass
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    ss = [Search(g, st) for g, st in zip(groups, streams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in ss:
        s.begin()
    more = True
    while more:
        more = False
        for s in ss:
            more = s.wave() or more
    torch.cuda.synchronize(); t_pair = time.perf_counter() - t0
    print(f"{W} in flight, {sims} simulations: one search of {R_all} roots {t_whole * 1e3:.1f} ms | {NG} of {R_all // NG} one after the other {t_seq * 1e3:.1f} ms | "
          f"alternating on {NG} queues {t_pair * 1e3:.1f} ms", flush=True)
