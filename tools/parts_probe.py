"""Fresh process: which hardware queues do the streams of the partitioned schedule share, and what does it cost?
ipp_probe_stream_pair over a pool of streams (+ the caller's stream) gives the same-queue classes (a pair on one queue
runs its two chains one after the other: 2x); then the step time of VecIPPEnv(parts=2) for part streams A, B and the
staging stream S taken from chosen classes.
    python tools/parts_probe.py"""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B, T = 4096, 40  # configs[1]
cfg = EngineConfig(x_dim=50, y_dim=50)
ALTS = [float(a) for a in range(5, 15)]
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=2)
env.reset()
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(T)]
pool = [torch.cuda.Stream() for _ in range(8)]
for s in pool:
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
torch.cuda.synchronize()
names = [f"p{i}" for i in range(len(pool))] + ["envA", "envB", "side", "main"]
allst = pool + list(env._part_streams) + [env._side, torch.cuda.current_stream()]
n = len(allst)
ms = np.zeros((n, n))
for i in range(n):
    for j in range(i + 1, n):
        ms[i, j] = ms[j, i] = min(env.engine.probe_stream_pair(allst[i], allst[j]) for _ in range(2))
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print(names)
print(ms)
lo = ms[ms > 0].min()
cls = [-1] * n
for i in range(n):
    if cls[i] < 0:
        cls[i] = max(cls) + 1
        for j in range(i + 1, n):
            if ms[i, j] > 1.5 * lo:
                cls[j] = cls[i]
print("queue classes:", dict(zip(names, cls)))


def run(tag, a, b, s, steps=300):
    env.wait(); torch.cuda.synchronize()
    env._part_streams = [allst[a], allst[b]]
    env._side = allst[s]
    env._main_dirty = True
    for t in range(40):
        env.step_async(acts[t % T], inputs_ready=True)
    env.wait(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for t in range(steps):
            env.step_async(acts[t % T], inputs_ready=True)
        env.wait(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"A={names[a]}(q{cls[a]}) B={names[b]}(q{cls[b]}) S={names[s]}(q{cls[s]}) main=q{cls[-1]}: {best * 1e6:.1f} us per step "
          f"({B / best / 1e6:.1f} M env-steps/s)  {tag}", flush=True)


print("the env's own pick:", env._queues and {k: v for k, v in env._queues.items() if k in ("n_queues",)})
run("(the env's own pick)", len(pool), len(pool) + 1, len(pool) + 2)
rep = {}
for i, c in enumerate(cls[:len(pool)]):
    rep.setdefault(c, i)
qs = sorted(rep)
print("representatives:", {q: names[rep[q]] for q in qs})
for qa, qb in itertools.combinations(qs, 2):
    for qsid in qs:
        run("", rep[qa], rep[qb], rep[qsid])
