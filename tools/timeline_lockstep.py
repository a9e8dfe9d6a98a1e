"""Per-item timeline (IPP_TIMELINE build, tools/probes/libipp_timing.so through IPP_HIP_LIB) of ONE launch of the headline's 4096 envs in lock step
at episode step k: k = 0 -> items without a stored column (the per-item fixed costs alone), k = 20 -> items of the staggered batch's mean rank.
    IPP_HIP_LIB=$PWD/tools/probes/libipp_timing.so python tools/timeline_lockstep.py <k> <dump.bin>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

k, path = int(sys.argv[1]), sys.argv[2]
B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
ALTS = [float(a) for a in range(5, 15)]
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=False, window_rows=-1, seed=1, parts=1)
for rep in range(2):  # (second episode: warm)
    env.reset()
    for t in range(k + 1):
        env.step(torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda"))
torch.cuda.synchronize()
os.environ["IPP_TIMELINE_FILE"] = path
env.engine.streamed_bytes()
print(f"lock step, episode step {k}: mean rank after the step {env.engine.ranks().float().mean().item():.1f}")
