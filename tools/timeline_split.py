"""Timeline of the SPLIT step in steady state (IPP_TIMELINE build, `make -C ipp-rl_amd/csrc timeline`): VecIPPEnv(parts=P) at
configs[1] with IPP_SPLIT=1, a few hundred steps, then the marks of the LAST step: the prologue kernel's phases per item, the unit
kernel's waves (start, tables in LDS, end), slot duty and residency over time.
    IPP_SPLIT=1 python tools/timeline_split.py [parts] [episode_steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPP_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libipp_timing.so"))
os.environ["IPP_TIMELINE_FILE"] = "/tmp/tl_split.bin"
os.environ.setdefault("IPP_SPLIT", "1")
import numpy as np
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 4096
cfg = EngineConfig(x_dim=50, y_dim=50)
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
env.engine.streamed_bytes(reset=True)  # (the timing build dumps the marks here)
t = np.fromfile("/tmp/tl_split.bin", dtype=np.uint64).reshape(-1, 8)[:B].astype(np.float64)
ut = np.fromfile("/tmp/tl_split.bin.units", dtype=np.uint64).reshape(-1, 2, 8, 4)[:B, 0].astype(np.float64)  # [pos][unit][item, start, tables, end]
have = (ut[:, :, 3] > 0) & (ut[:, :, 1] >= t[t[:, 2] > 0, 0].min())  # (units of the last step only: older steps' marks of higher unit indices stay in the trace)
t0 = t[t[:, 2] > 0, 0].min()
us = lambda x: (x - t0) / 100.0
# ---- prologue kernel: marks 0 start, 3 header, 4 compaction, 5 tables, 1 gather done, 7 algebra done, 2 block written
names = [("header", 0, 3), ("columns", 3, 4), ("gather issue + tables", 4, 5), ("gather", 5, 1), ("algebra (+ observation)", 1, 7), ("block", 7, 2)]
ok = t[:, 2] > 0
print(f"parts {env.parts}, split waves {os.environ.get('IPP_SPLIT_WAVES', 'default')}; prologue kernel, {int(ok.sum())} items of the last step:")
print(f"  item duration mean {np.mean(us(t[ok, 2]) - us(t[ok, 0])):.1f} us  p50 {np.percentile(t[ok, 2] - t[ok, 0], 50) / 100:.1f}  p90 {np.percentile(t[ok, 2] - t[ok, 0], 90) / 100:.1f}  max {np.max(t[ok, 2] - t[ok, 0]) / 100:.1f}")
for nm, a, b in names:
    d = (t[ok, b] - t[ok, a]) / 100.0
    print(f"    {nm:26s} mean {d.mean():5.1f}  p90 {np.percentile(d, 90):5.1f}  max {d.max():5.1f}")
ps, pe = us(t[ok, 0]), us(t[ok, 2])
print(f"  launch span: first start {ps.min():.1f}, last start {ps.max():.1f}, last end {pe.max():.1f} us")
# ---- unit kernel
s_, m_, e_ = us(ut[:, :, 1][have]), us(ut[:, :, 2][have]), us(ut[:, :, 3][have])
print(f"unit kernel: {int(have.sum())} units ({have.sum() / ok.sum():.2f} per item): duration mean {np.mean(e_ - s_):.1f} us  p50 {np.percentile(e_ - s_, 50):.1f}  p90 {np.percentile(e_ - s_, 90):.1f}  max {np.max(e_ - s_):.1f};"
      f"  tables -> LDS mean {np.mean(m_ - s_):.2f} us  p90 {np.percentile(m_ - s_, 90):.2f}")
print(f"  slot duty (stream / held) = {np.sum(e_ - m_) / np.sum(e_ - s_):.3f};  first start {s_.min():.1f}, last start {s_.max():.1f}, last end {e_.max():.1f} us")
bucket = 5.0
print(" t [us]  prologue items resident   unit waves resident")
for a in np.arange(0.0, max(pe.max(), e_.max()) + bucket, bucket):
    print(f"  {a:5.0f}   {int(((ps <= a) & (pe > a)).sum()):6d}   {int(((s_ <= a) & (e_ > a)).sum()):6d}")
