#!/usr/bin/env python3
"""Per-phase wave clocks of the patch step kernel (a -DIPP_TIMELINE=1 -DIPP_WAVE_CLOCKS=0x1ff build: proportions only, the clock reads
slow the kernel several times).  usage: IPP_HIP_LIB=tools/probes/libipp_wt.so python tools/wave_phases.py [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1)
alts = [float(a) for a in range(5, 15)]
env.reset()
for t in range(T + 4):
    env.step(cell_centre_actions(cfg, t, 0, B, B, alts))
torch.cuda.synchronize()
env.engine.streamed_bytes_detail(reset=True)  # (prints and clears the phase sums)
for t in range(T + 4, T + 12):
    env.step(cell_centre_actions(cfg, t, 0, B, B, alts))
torch.cuda.synchronize()
print("8 steps:", file=sys.stderr)
env.engine.streamed_bytes_detail(reset=True)
