#!/usr/bin/env python3
"""
Interleaved A/B timing of engine builds (different libipp_*.so variants) inside ONE process: every variant gets
its own engine with identical state and inputs (staggered 40-step episodes, cfg2 workload) and the variants take
turns step by step, so clock / thermal drift hits all of them equally.  Reports the median kernel times from the
engines' own HIP-event profiling and the median wall time per step.

usage: python tools/ab_kernels.py [--envs 4096] [--rounds 60] name=path/to/lib.so[:tile_threads] ...
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ipp_rl_amd import _ffi, EngineConfig  # noqa: E402
from ipp_rl_amd.engine import IPPEngine  # noqa: E402
from ipp_rl_amd.vec_env import cell_centre_actions  # noqa: E402


def make_engine(lib_path, cfg, B, tile_threads, window_rows=0):
    _ffi._lib = None
    _ffi.LIB_PATH = lib_path
    eng = IPPEngine(cfg, capacity=B, state="factor", rank_cap=360, tile_threads=tile_threads, window_rows=window_rows,
                    fixed_prior=(0 < window_rows < 12))
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--window-rows", type=int, default=12)
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--episode-steps", type=int, default=40)
    ap.add_argument("--order", default="natural", help="comma list per variant: natural | desc | asc (items sorted by rank)")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    cfg = EngineConfig(x_dim=args.grid, y_dim=args.grid)
    B, T = args.envs, args.episode_steps
    alts = [float(a) for a in range(5, 15)]
    n_steps = T + args.rounds
    acts = torch.stack([torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, alts)) for t in range(n_steps)]).cuda()
    init = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device="cuda").repeat(B, 1)
    phase = torch.arange(B, device="cuda") % T
    ids_by_phase = [torch.nonzero(phase == p).flatten().to(torch.int32) for p in range(T)]
    gt = torch.rand((B, cfg.n_cells), device="cuda")
    noise = torch.randn((n_steps, B, 9), device="cuda")
    engines = []
    for spec in args.variants:
        name, rest = spec.split("=")
        path, _, tt = rest.partition(":")
        eng = make_engine(os.path.abspath(path), cfg, B, int(tt or 0), args.window_rows)
        eng.reset(gt=gt)
        engines.append((name, eng, init.clone()))
    reward = torch.empty(B, dtype=torch.float32, device="cuda")
    status = torch.empty(B, dtype=torch.int32, device="cuda")
    flags = _ffi.IPP_ADAPTIVE | _ffi.IPP_USE_FLIGHT_TIME
    wall = {n: [] for n, _, _ in engines}

    orders = (args.order.split(",") * len(engines))[: len(engines)]
    order_of = {n: o for (n, _, _), o in zip(engines, orders)}

    def one(eng, prev, t, timed_name=None):
        ids_sorted = a_t = p_t = n_t = None
        mode = order_of[timed_name] if timed_name else "natural"
        if mode != "natural":  # item i works on env perm[i]; inputs are per item
            perm = torch.argsort(eng.ranks(), descending=(mode != "asc"), stable=True)
            if mode == "zigzag":  # heaviest, lightest, 2nd heaviest, 2nd lightest, ...: mixed load, medium items last
                half = (B + 1) // 2
                z = torch.empty_like(perm)
                z[0::2] = perm[:half]
                z[1::2] = perm.flip(0)[: B - half]
                perm = z
            elif mode.startswith("spread"):  # heaviest 1/k at the head of each of the 16 ranges of the XCD map, rest natural
                k = int(mode[6:] or 16)
                nh = B // k
                head = perm[:nh]
                keep = torch.ones(B, dtype=torch.bool, device=perm.device)
                keep[head] = False
                rest = torch.nonzero(keep).flatten()
                ln = B // 16
                hp, rp = nh // 16, ln - nh // 16
                parts = []
                for q in range(16):
                    parts.append(head[q * hp:(q + 1) * hp])
                    parts.append(rest[q * rp:(q + 1) * rp])
                perm = torch.cat(parts)
            elif mode.startswith("head"):  # "head<k>[tail<j>]": heaviest 1/k first, lightest 1/j last, the rest natural
                k, _, j = mode[4:].partition("tail")
                head = perm[: B // int(k)]
                tail = perm[B - B // int(j):].flip(0) if j else perm[:0]
                keep = torch.ones(B, dtype=torch.bool, device=perm.device)
                keep[head] = False
                keep[tail] = False
                perm = torch.cat([head, torch.nonzero(keep).flatten(), tail])
            ids_sorted = perm.to(torch.int32)
            a_t, p_t, n_t = acts[t][perm].contiguous(), prev[perm].contiguous(), noise[t][perm].contiguous()
        if timed_name:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if mode == "natural":
            eng.step_raw(B, acts[t], prev, noise[t], flags, reward, status)
        else:
            eng.step_raw(B, a_t, p_t, n_t, flags, reward, status, env_ids=ids_sorted)
        if timed_name:
            torch.cuda.synchronize()
            wall[timed_name].append(time.perf_counter() - t0)
        prev.copy_(acts[t])
        p = (T - ((t + 1) % T)) % T
        ids = ids_by_phase[p]
        eng.reset(env_ids=ids, gt=gt[: ids.numel()])
        prev[ids.long()] = init[ids.long()]

    for t in range(T):  # pre-roll to the stationary rank mix
        for name, eng, prev in engines:
            one(eng, prev, t)
    for _, eng, _ in engines:
        eng.profile(True)
    for r in range(args.rounds):
        order = engines if r % 2 == 0 else engines[::-1]
        for name, eng, prev in order:
            one(eng, prev, T + r, timed_name=name)
    torch.cuda.synchronize()
    if os.environ.get("IPP_TIMELINE_FILE"):  # timing builds dump the last launch's timeline here
        engines[0][1].streamed_bytes()
    base = None
    for name, eng, _ in engines:
        g_ms, g_n = eng.profile_read(0)
        p_ms, _ = eng.profile_read(2)
        w = 1e3 * float(np.median(wall[name]))
        ranks = eng.ranks().double().mean().item()
        base = base or g_ms
        print(f"{name:14s} T={eng.info.tile_threads:3d} gain {g_ms:.4f} ms ({g_ms / base:5.3f}x)  prepare {p_ms:.4f} ms  "
              f"step wall median {w:.4f} ms  mean rank {ranks:.1f}  order {order_of[name]}")


if __name__ == "__main__":
    main()
