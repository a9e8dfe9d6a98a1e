#!/usr/bin/env python3
"""
Accuracy and bytes of the windowed factor state as a function of window_rows, on the 40-step 50x50 golden episodes
recorded from the reference (tests/golden/episode_*_50_*.npz): worst |error| of reward / mean / diag over the
episode and of the sampled rows of the final P, and the streamed bytes relative to full columns.

usage: python tools/window_sweep.py [--fixed-prior] [window_rows ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ipp_rl_amd import EngineConfig, IPPEngine  # noqa: E402


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def main():
    fixed = "--fixed-prior" in sys.argv[1:]  # window bound for l itself instead of 1.2 l (no shuffle_prior_cov)
    windows = [int(a) for a in sys.argv[1:] if a != "--fixed-prior"] or [6, 7, 8, 9, 10, 11, 12, 14, 0]
    for name in ("episode_rf1_50_s0", "episode_mixed_50_s1"):
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        dim = g["gt"].shape[0]
        for w in windows:
            cfg = EngineConfig(x_dim=dim, y_dim=dim)
            try:
                eng = IPPEngine(cfg, capacity=2, state="factor", rank_cap=360, window_rows=w, fixed_prior=fixed)
            except Exception as exc:  # window refused by the engine's own bound
                print(f"{name} window {w:2d}: refused ({str(exc)[:60]})")
                continue
            eng.reset(env_ids=[0], white_noise=g["white"][None])
            prev = np.array([2.0, 2.0, 14.0])
            worst = dict(reward=0.0, mean=0.0, diag=0.0)
            eng.streamed_bytes(reset=True)
            for t, a in enumerate(g["actions"]):
                reward, status = eng.step(a[None], prev[None], env_ids=[0], meas_noise=g["eps"][t][None])
                worst["reward"] = max(worst["reward"], abs(float(reward[0]) - g["reward"][t]))
                worst["mean"] = max(worst["mean"], np.max(np.abs(host(eng.read_mean(0)) - g["mean"][t])))
                worst["diag"] = max(worst["diag"], np.max(np.abs(host(eng.read_diag(0)) - g["diag"][t])))
                prev = a
            streamed = eng.streamed_bytes()
            N = dim * dim
            r_before = np.concatenate([[0], np.cumsum(g["m"])[:-1]])
            full = float(np.sum(4.0 * N * (r_before + g["m"]) + 16.0 * N))
            err_rows = np.max(np.abs(host(eng.read_cov(0))[g["sample_rows"]] - g["P_final_rows"]))
            print(f"{name} window {w:2d}: reward {worst['reward']:.1e} mean {worst['mean']:.1e} diag {worst['diag']:.1e} "
                  f"P rows {err_rows:.1e}  streamed {streamed / full:.3f} of full columns")
            eng.close()


if __name__ == "__main__":
    main()
