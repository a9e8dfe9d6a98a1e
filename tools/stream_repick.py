"""Is the fast / slow mode of a large batch a property of the STREAMS an env picked or of its memory?  One env (configs[3] share), the
part streams re-picked several times, the same steps timed after every pick.  usage: python tools/stream_repick.py [envs] [steps] [picks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
picks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
T = 40
cfg = EngineConfig(x_dim=50, y_dim=50)
ALTS = [float(a) for a in range(5, 15)]
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(T + 4 * steps)]
for e in range(2):
    env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=2)
    env.reset()
    t = 0
    for _ in range(T + steps):
        env.step_async(acts[t], inputs_ready=True); t += 1
    env.wait(); torch.cuda.synchronize()
    for k in range(picks):
        if k:
            env.wait(); torch.cuda.synchronize()
            env._part_streams, side = env._pick_streams(env.device, 2)
            env._main_dirty = True
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            for i in range(steps):
                env.step_async(acts[T + steps + i], inputs_ready=True)
            env.wait(); torch.cuda.synchronize()
            times.append(1e3 * (time.perf_counter() - t0) / steps)
        print(f"env {e} pick {k}: streams {[hex(s.cuda_stream)[-6:] for s in env._part_streams]}  ms per step {[round(x, 4) for x in times]}")
    env.close(); del env
    torch.cuda.empty_cache()
