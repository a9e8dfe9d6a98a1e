#!/usr/bin/env python3
"""Host time per VecIPPEnv step call against the device time per step (is the loop launch-bound?).  usage: python tools/host_overhead.py [parts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, T, K = int(os.environ.get("ENVS", 4096)), 40, 200
cfg = EngineConfig(x_dim=50, y_dim=50)
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, parts=parts)
alts = [float(a) for a in range(5, 15)]
acts = torch.stack([torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts)) for t in range(T + 2 * K + 8)]).cuda()
env.reset()
for t in range(T + 8):
    env.step(acts[t])
torch.cuda.synchronize()
for mode in ("async", "sync"):
    if mode == "async" and env.parts <= 1:
        continue
    t0 = time.perf_counter()
    for t in range(K):
        if mode == "async":
            env.step_async(acts[T + 8 + t], inputs_ready=True)
        else:
            env.step(acts[T + 8 + K + t], after_step_hook=(lambda: None) if env.parts > 1 else None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"parts {env.parts} {mode}: host issue {1e6 * (t1 - t0) / K:.1f} us/step, until the device is idle {1e6 * (t2 - t0) / K:.1f} us/step")
