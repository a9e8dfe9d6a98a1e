#!/usr/bin/env python3
"""Throughput of ipp_tree_step: B root envs, one simulation per root at a time, each descending `depth` levels
(expand at every level), like one wave of MCTS simulations (planning/mcts_zero/mcts.py:166-265).
usage: python tools/tree_bench.py [--grid 50] [--roots 4096] [--depth 5] [--root-steps 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipp_rl_amd import EngineConfig, IPPEngine  # noqa: E402
from ipp_rl_amd.vec_env import cell_centre_actions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--roots", type=int, default=4096)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--root-steps", type=int, default=10)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    cfg = EngineConfig(x_dim=args.grid, y_dim=args.grid)
    B, Dp = args.roots, args.depth
    alts = [float(a) for a in range(5, 15)]
    eng = IPPEngine(cfg, capacity=B, state="factor", rank_cap=9 * (args.root_steps + Dp + 1), window_rows=12, node_capacity=B * Dp)
    white = eng.normal(B * cfg.n_cells, seed=3).reshape(B, -1)
    eng.reset(white_noise=white)
    prev = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(B, 1)
    for t in range(args.root_steps):
        a = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts), device="cuda")
        eng.step(a, prev, meas_noise=eng.normal(B * 9, seed=50 + t).reshape(B, 9))
        prev = a
    roots = torch.arange(B, dtype=torch.int32, device="cuda")
    acts = [torch.as_tensor(cell_centre_actions(cfg, 100 + d, 0, B, B, alts), device="cuda") for d in range(Dp)]
    paths = torch.full((Dp, B, 6), -1, dtype=torch.int32, device="cuda")
    new_ids = [(d * B + torch.arange(B, device="cuda")).to(torch.int32) for d in range(Dp)]
    for d in range(1, Dp):
        paths[d] = paths[d - 1]
        paths[d, :, d - 1] = new_ids[d - 1]
    reward = torch.empty(B, dtype=torch.float32, device="cuda")
    status = torch.empty(B, dtype=torch.int32, device="cuda")

    def one_wave(keep=None):
        p = prev
        for d in range(Dp):
            eng.tree_step(roots, paths[d], acts[d], p, new_ids=new_ids[d], reward_out=reward, status_out=status)
            if keep is not None:
                keep.append(reward.clone())
            p = acts[d]

    one_wave(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        one_wave()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.reps * 1e3
    assert int(status.abs().sum()) == 0
    # the same wave again and again must give the same bits (a data race in the kernel would not): 40 waves
    ref = []
    one_wave(ref)
    mismatches = 0
    for _ in range(40):
        got = []
        one_wave(got)
        mismatches += sum(int((a != b).sum()) for a, b in zip(ref, got))
    assert mismatches == 0, f"{mismatches} rewards differ between identical simulation waves"
    print(f"[{args.grid}x{args.grid}, {B} roots, root rank {float(eng.ranks().float().mean()):.0f}, depth {Dp}] "
          f"{ms:.3f} ms per simulation wave = {B * Dp / ms * 1e3:.3e} tree steps/s ({ms / Dp:.3f} ms per level)")


if __name__ == "__main__":
    main()
