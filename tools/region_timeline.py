"""Where the time of a SHORT timed region goes (the driver's protocol: 20 steps between device syncs): configs[1] on the two-group
schedule with a HIP event pair on every dispatch (ipp_profile_enable + IPP_PROFILE_DUMP), one region of `steps` steps from an idle
device; prints every launch's start / stop, the wall time of each step (start of its first launch to the start of the next step's)
and the fill / steady / drain account.   usage: python tools/region_timeline.py [steps] [parts]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DUMP = "/tmp/ipp_prof_dump.txt"
os.environ["IPP_PROFILE_DUMP"] = DUMP
import numpy as np
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=parts)
env.reset()
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, [float(a) for a in range(5, 15)]), device="cuda") for t in range(T + 3 * steps)]
t = 0
for _ in range(T + steps):
    (env.step_async(acts[t], inputs_ready=True) if env.parts > 1 else env.step(acts[t])); t += 1
env.wait(); torch.cuda.synchronize()
if os.path.exists(DUMP):
    os.remove(DUMP)
env.engine.profile(True)
t0 = time.perf_counter()
for _ in range(steps):
    (env.step_async(acts[t], inputs_ready=True) if env.parts > 1 else env.step(acts[t])); t += 1
env.wait(); torch.cuda.synchronize()
wall = 1e6 * (time.perf_counter() - t0)
env.engine.profile_read(0)
env.engine.profile(False)
rows = np.array([[float(x) for x in l.split()] for l in open(DUMP) if not l.startswith("#")])
rows = rows[np.argsort(rows[:, 1])]
n_l = len(rows)
per_step = n_l // steps
print(f"{steps} steps, {parts} group(s): host wall {wall:.0f} us = {wall / steps:.1f} us per step (event pairs on every dispatch); {n_l} dispatches, {per_step} per step")
starts = rows[:, 1].reshape(steps, per_step)[:, 0]
ends = rows[:, 2].reshape(steps, per_step).max(axis=1)
first, last = rows[0, 1], rows[:, 2].max()
print(f"device span first start -> last stop: {last - first:.1f} us = {(last - first) / steps:.1f} us per step")
d = np.diff(starts)
print("step  start_us  stop_us  until the next step starts")
for i in range(steps):
    print(f"{i:4d}  {starts[i] - first:8.1f} {ends[i] - first:8.1f}  {d[i] if i < steps - 1 else float('nan'):6.1f}")
mid = d[steps // 3: -2] if steps >= 12 else d
print(f"steady step period (middle of the region) {np.median(mid):.1f} us; the first {min(4, steps - 1)} periods {np.round(d[:4], 1).tolist()}; "
      f"drain (last stop - last start) {ends[-1] - starts[-1]:.1f} us; fill + drain over the steady rate: {(last - first) - steps * np.median(mid):.1f} us per region")
