"""
Stress for data races in the step kernels: four engine paths (fused window 10 / 12, workgroup-per-item kernel,
exact mode) step the same 4096 staggered envs with the same seeds; any env whose reward leaves the median of the four
by more than 1e-5 is reported.  Found the missing vmcnt wait in front of the fused kernel's scalar read-back of the
-HT rows (one wrong env in ~10^5 item steps).  usage: [DBG_REPS=12] python tools/race_stress.py
"""
import sys, os
sys.path.insert(0, os.environ.get("DBG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPP_POISON_ARENA", "1")
import torch, numpy as np
from ipp_rl_amd import EngineConfig, IPPEngine
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

cfg = EngineConfig(x_dim=50, y_dim=50)
B, T, steps = 4096, 40, 50
alts = [float(a) for a in range(5, 15)]
junk = []
for rep in range(int(os.environ.get("DBG_REPS", "6"))):
    junk = [torch.full((np.random.randint(1, 50) * 1000003,), float("nan"), device="cuda") for _ in range(6)]
    junk = []
    envs = {"fused10": VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=(14 if os.environ.get("DBG_ROOT") else -1), seed=77),
            "exact": VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=0, seed=77),
            "t128w12": VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=12, tile_threads=128, seed=77),
            "fused12": VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=12, seed=77)}
    for e in envs.values():
        e.reset()
    for t in range(steps):
        acts = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts)).cuda()
        r = {k: e.step(acts)[0].double().clone() for k, e in envs.items()}
        med = torch.stack(list(r.values())).median(dim=0).values
        for k in r:
            d = (r[k] - med).abs()
            bad = torch.nonzero(d > 1e-5).flatten()
            if bad.numel():
                e = int(bad[0])
                print(f"rep {rep} step {t}: {k} deviates from the median in {bad.numel()} envs, first env {e} (phase {e % T}): "
                      + " ".join(f"{kk}={float(r[kk][e]):.6f}" for kk in r))
    print(f"rep {rep} done")
    del envs
