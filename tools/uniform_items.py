"""How much of a step is the spread of item weights?  The headline's batch holds every episode phase (ranks 0 .. 351 in one launch: the launch lasts as
long as its heaviest items); the same 4096 envs in LOCK STEP hold one phase per step, so step k is a launch of items of (nearly) one weight -- mean
rank 5.9 k.  One launch per step both times; per-step HIP-event durations of the lock-step env against the staggered env's average.
    python tools/uniform_items.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
ALTS = [float(a) for a in range(5, 15)]
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(4 * T)]

def run(stagger, reps=3):
    env = VecIPPEnv(cfg, B, episode_steps=T, stagger=stagger, window_rows=-1, seed=1, parts=1)
    env.reset()
    per_step = np.zeros((reps, T))
    t = 0
    for _ in range(T):  # warm-up episode
        env.step(acts[t % len(acts)]); t += 1
    if not stagger:
        env.reset()
    for rep in range(reps):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(T)]
        for k in range(T):
            evs[k][0].record()
            env.step(acts[t % len(acts)]); t += 1
            evs[k][1].record()
        torch.cuda.synchronize()
        per_step[rep] = [a.elapsed_time(b) for a, b in evs]
        if not stagger:
            env.reset()
    ranks = env.engine.ranks().float().mean().item()
    env.close()
    return per_step.min(axis=0)

stag = run(True)
lock = run(False)
print(f"staggered (every phase in every launch): {1e3 * stag.mean():.1f} us per step (min over reps per step, mean over 40 steps; includes the event pair and host gaps)")
print("lock step, step k of the episode (mean rank 5.9 k before the step): us")
for k0 in range(0, T, 8):
    print("   " + "  ".join(f"k={k:2d}: {1e3 * lock[k]:5.1f}" for k in range(k0, min(T, k0 + 8))))
print(f"lock step, mean over the episode: {1e3 * lock.mean():.1f} us;  at k = 20 (the staggered launch's mean rank): {1e3 * lock[18:23].mean():.1f} us")
