"""Per-item timeline of the partitioned schedule in steady state (IPP_TIMELINE build, `make -C ipp-rl_amd/csrc timeline`):
VecIPPEnv(parts=P) at configs[1], a few hundred async steps, then the marks of every item's LAST step (100 MHz wall
clock): item durations, phases, and how many workgroup slots are taken over time.
    IPP_HIP_LIB=tools/probes/libipp_timing.so python tools/timeline_parts.py [parts] [episode_steps] [grid] [envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPP_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libipp_timing.so"))
os.environ["IPP_TIMELINE_FILE"] = "/tmp/tl_parts.bin"
import numpy as np
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
G = int(sys.argv[3]) if len(sys.argv) > 3 else 50
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
cfg = EngineConfig(x_dim=G, y_dim=G)
ALTS = [float(a) for a in range(5, 15)]
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=P)
env.reset()
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(4 * T)]
n_steps = 3 * T + 7
for t in range(n_steps):
    if env.parts > 1:
        env.step_async(acts[t % len(acts)], inputs_ready=True)
    else:
        env.step(acts[t % len(acts)])
if env.parts > 1:
    env.wait()
torch.cuda.synchronize()
ranks = env.engine.ranks().cpu().numpy()
env.engine.streamed_bytes(reset=True)  # (the timing build dumps the marks here)
t = np.fromfile("/tmp/tl_parts.bin", dtype=np.uint64).reshape(-1, 8)[:B].astype(np.float64)
t0 = t[:, 0].min()
start, mid, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
dur = end - start
grp = np.zeros(B, dtype=np.int64)
if env.parts > 1:
    for g in range(env.parts):
        grp[env.part_envs(g).cpu().numpy()] = g
print(f"parts {env.parts}, {B} items, last step: span {end.max():.1f} us")
for g in range(env.parts):
    m = grp == g
    print(f" group {g}: {int(m.sum())} items, launch from {start[m].min():.1f} to {end[m].max():.1f} us ({end[m].max() - start[m].min():.1f}), "
          f"last start {start[m].max():.1f}")
print(f"item duration: mean {dur.mean():.1f} us  p10 {np.percentile(dur, 10):.1f}  p50 {np.percentile(dur, 50):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f}"
      f"   sum / 2048 slots = {dur.sum() / 2048:.1f} us")
print(f"phase A (prologue): mean {np.mean(mid - start):.1f}  p90 {np.percentile(mid - start, 90):.1f};  phase B (units): mean {np.mean(end - mid):.1f}  p90 {np.percentile(end - mid, 90):.1f}")
# by weight (the rank the item had when it ran = steps done in its episode x 9): duration against rank
w = ((n_steps - 1 + (np.arange(B) % T)) % T)
for lo in range(0, T, max(1, T // 8)):
    m = (w >= lo) & (w < lo + max(1, T // 8))
    print(f"  episode step {lo:2d}-{lo + max(1, T // 8) - 1:2d}: n {int(m.sum()):4d}  duration mean {dur[m].mean():5.1f}  phase A {np.mean((mid - start)[m]):5.1f}  phase B {np.mean((end - mid)[m]):5.1f}  start mean {start[m].mean():5.1f}")
alt = acts[(n_steps - 1) % len(acts)][:, 2].cpu().numpy()
print(f"duration percentiles: p99 {np.percentile(dur, 99):.1f}  p99.5 {np.percentile(dur, 99.5):.1f}  p99.9 {np.percentile(dur, 99.9):.1f}")
print(" longest items:  env  group  episode step  altitude  start  phase A  phase B  duration")
for i in np.argsort(-dur)[:16]:
    print(f"   {i:5d}  {grp[i]}  {w[i]:3d}  {alt[i]:4.0f}  {start[i]:6.1f}  {mid[i] - start[i]:5.1f}  {end[i] - mid[i]:5.1f}  {dur[i]:5.1f}")
print(" phase B by altitude for episode steps >= 30:")
for a in sorted(set(alt.tolist())):
    m = (alt == a) & (w >= 30)
    if m.sum():
        print(f"   altitude {a:4.0f}: n {int(m.sum()):3d}  phase B mean {np.mean((end - mid)[m]):5.1f}  max {np.max((end - mid)[m]):5.1f}   duration mean {dur[m].mean():5.1f} max {dur[m].max():5.1f}")
bucket = 5.0
edges = np.arange(0.0, end.max() + bucket, bucket)
print(" t [us]  resident  in prologue  (per group resident)")
for a in edges:
    res = (start <= a) & (end > a)
    pro = (start <= a) & (mid > a)
    print(f"  {a:5.0f}   {int(res.sum()):5d}   {int(pro.sum()):5d}   " + " ".join(str(int((res & (grp == g)).sum())) for g in range(env.parts)))
