"""Is a 20-step timed region slower than 20 steps INSIDE a long region?  configs[1], two groups: (a) regions of `k` steps bracketed
by device syncs (bench.py's protocol), (b) one long region with a timing event on every part stream every `k` steps.
usage: python tools/region_chunks.py [k] [chunks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
idle_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=2)
env.reset()
n = T + 2 * k * chunks + 8
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, [float(a) for a in range(5, 15)]), device="cuda") for t in range(n)]
t = 0
for _ in range(T):
    env.step_async(acts[t], inputs_ready=True); t += 1
env.wait(); torch.cuda.synchronize()
# (a) synced regions
reg = []
for c in range(chunks):
    torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms * 1e-3)
    t0 = time.perf_counter()
    for _ in range(k):
        env.step_async(acts[t], inputs_ready=True); t += 1
    env.wait(); torch.cuda.synchronize()
    reg.append(1e6 * (time.perf_counter() - t0) / k)
# (b) one long region, events every k steps on both part streams
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)] for _ in range(2)]
torch.cuda.synchronize()
for p in range(2):
    evs[p][0].record(env.part_stream(p))
for c in range(chunks):
    for _ in range(k):
        env.step_async(acts[t], inputs_ready=True); t += 1
    for p in range(2):
        evs[p][c + 1].record(env.part_stream(p))
env.wait(); torch.cuda.synchronize()
inside = [max(evs[p][c].elapsed_time(evs[p][c + 1]) for p in range(2)) * 1e3 / k for c in range(chunks)]
print(f"{k}-step regions between syncs (idle {idle_ms} ms before each): us per step {np.round(reg, 1).tolist()}  median {np.median(reg):.1f}")
print(f"{k}-step chunks inside one region of {k * chunks} steps:        us per step {np.round(inside, 1).tolist()}  median {np.median(inside):.1f}")
