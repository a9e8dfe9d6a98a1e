"""Host time of every VecIPPEnv.step_async call at configs[1] (two groups): which steps carry the staging of a ground-truth block
or a noise-ring refill, and what they cost on the host -- a step whose issue takes longer than the device needs for the queued
steps is a bubble.  usage: python tools/host_step_times.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=2)
env.reset()
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, [float(a) for a in range(5, 15)]), device="cuda") for t in range(T + n)]
for t in range(T):
    env.step_async(acts[t], inputs_ready=True)
env.wait(); torch.cuda.synchronize()
ts = []
t_all = time.perf_counter()
for t in range(T, T + n):
    t0 = time.perf_counter()
    env.step_async(acts[t], inputs_ready=True)
    ts.append(1e6 * (time.perf_counter() - t0))
issue = time.perf_counter() - t_all
env.wait(); torch.cuda.synchronize()
total = time.perf_counter() - t_all
ts = np.array(ts)
print(f"{n} steps: issue {1e6 * issue / n:.1f} us per step on the host, device {1e6 * total / n:.1f} us per step; host per call: median {np.median(ts):.1f}  p90 {np.percentile(ts, 90):.1f}  max {ts.max():.1f}")
K = env._blk_K
print("step  host_us   (B = first step of a staged block of %d, R = noise ring refill)" % K)
for i, x in enumerate(ts):
    t = T + i
    tag = ("B" if t % K == 0 else " ") + ("R" if t % env.NOISE_RING == 0 else " ")
    if x > 1.5 * np.median(ts) or tag.strip():
        print(f"{t:4d}  {x:7.1f}  {tag}")
