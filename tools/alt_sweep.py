"""How much of a step is its stragglers?  The two-group schedule at configs[1] with the altitude levels restricted (the longest
items are the altitude-13/14 footprints): us per step and counted bytes per step, so that the time per byte can be compared.
    python tools/alt_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
for name, alts in (("5-14 (bench)", range(5, 15)), ("5-12", range(5, 13)), ("5-10", range(5, 11)), ("9-10", range(9, 11)), ("13-14", range(13, 15))):
    alts = [float(a) for a in alts]
    for parts in (2, 1):
        env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=parts)
        env.reset()
        acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts), device="cuda") for t in range(4 * T)]

        def run(n):
            for t in range(n):
                if env.parts > 1:
                    env.step_async(acts[t % len(acts)], inputs_ready=True)
                else:
                    env.step(acts[t % len(acts)])
            if env.parts > 1:
                env.wait()
            torch.cuda.synchronize()

        run(2 * T)
        env.engine.streamed_bytes(reset=True)
        t0 = time.perf_counter()
        run(200)
        dt = (time.perf_counter() - t0) / 200
        counted, _ = env.engine.streamed_bytes_detail(reset=True)
        mb = counted / 200 / 1e6
        print(f"altitudes {name:14s} parts {parts}: {dt * 1e6:6.1f} us per step, {mb:6.1f} MB counted per step -> {mb / (dt * 1e6) * 1e6 / 8e6:.3f} of 8 TB/s; "
              f"{B / dt / 1e6:.1f} M env-steps/s", flush=True)
        env.engine.close()
        del env
