"""Step time of the two-group schedule with EXTRA vector instructions in every unit (the -DIPP_EXIT_POINTS=1 build; results
unchanged): does the time follow the instruction count (tools/valu_sections.sh), i.e. is the step bound by vector issue?
    IPP_HIP_LIB=tools/probes/libipp_issue.so python tools/skip_timing.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
alts = [float(a) for a in range(5, 15)]
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, parts=2)
env.reset()
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts), device="cuda") for t in range(4 * T)]


def run(tag, point, steps=400):
    env.wait(); torch.cuda.synchronize()
    env.engine._lib.ipp_debug_capture(env.engine._h, 1 + point if point else 0)  # (1 + point >= 11: extra instructions)
    for t in range(T):
        env.step_async(acts[t % len(acts)], inputs_ready=True)
    env.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        env.step_async(acts[t % len(acts)], inputs_ready=True)
    env.wait(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    bad = int((env.status != 0).sum())
    print(f"{tag:44s} {dt * 1e6:6.1f} us per step   (items with non-zero status: {bad}, mean rank {float(env.engine.ranks().float().mean()):.1f})", flush=True)


run("whole kernel", 0)
run("+128 vector instructions per unit (+2.0 M, +6 %)", 10)
run("+256 vector instructions per unit (+4.0 M, +12 %)", 11)
run("+512 vector instructions per unit (+8.0 M, +25 %)", 13)
run("whole kernel", 0)
