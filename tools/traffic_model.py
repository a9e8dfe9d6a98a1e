#!/usr/bin/env python3
"""
Structural HBM traffic of one env step of the headline workload (BASELINE configs[1]) by ADDRESS ARITHMETIC: the rectangles of
every env's stored columns are a function of the synthetic action schedule (cell_centre_actions), so the lines a launch must touch
can be counted on the host -- for the patch layout as built and for candidate layouts -- and set against the PMC counters.

For every item of one steady-state step: the new rectangle, its units (64 lanes x 2 cells over the rectangle's own width), the
contributing stored steps (rectangle meets the footprint), and per (unit, stored column) the 8-byte lane requests of the row stream:
    useful bytes           lanes inside the stored column's rectangle x 8           (= roofline.necessary_bytes' row part)
    lines_128 / sectors_64 distinct 128-byte lines / 64-byte sectors of that ONE wave request (no sharing between requests counted)
plus the column-overlap histogram the request shapes come from, the gather (one 4-byte load per footprint cell and contributing
column), the units' mean / variance reads and the writes.  No GPU, no engine: pure NumPy.
    python tools/traffic_model.py [--envs 4096] [--step 200] [--pw 26] [--layout row|half|tile4x8]
"""
import argparse
import collections
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def footprint(W, H, res, tanx, tany, rf_alt, a):
    """(xl, xr, yu, yd, rf) of action a = (x, y, z): sensors/cameras.py:34-75,122-125 as csrc/k_prepare.h make_item_header."""
    cx, cy = math.floor(2 * a[2] * tanx / res), math.floor(2 * a[2] * tany / res)
    gx, gy = math.floor(a[0] / res), math.floor(a[1] / res)
    rx, ry = math.floor(0.5 * cx), math.floor(0.5 * cy)
    clip = lambda v, hi: int(min(max(v, 0), hi))
    return clip(gx - rx, W - 1), clip(gx + rx, W - 1), clip(gy - ry, H - 1), clip(gy + ry, H - 1), (2 if a[2] > rf_alt else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--episode-steps", type=int, default=40)
    ap.add_argument("--step", type=int, default=200, help="bench step index that is modelled (steady state: >= episode steps)")
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--layout", default="row", help="row: patch rows of pw floats (as built); half: two half-width patches; tile4x8: 4 x 8-cell tiles of 128 B")
    ap.add_argument("--pw", type=int, default=0, help="row stride of a patch in floats (0: the engine's 2 R + 6)")
    ap.add_argument("--vec", type=int, default=2, help="cells per lane: 2 = as built (8-byte lanes, rectangles on even columns, units of 128 cells); 4 = the "
                                                       "16-byte-lane unit of VERDICT r05 item 2 (rectangles on multiples of 4 columns -- virtual columns beyond "
                                                       "an odd grid edge --, units of 256 cells, pw rounded up to a multiple of 4)")
    args = ap.parse_args()
    from ipp_rl_amd.engine import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions

    cfg = EngineConfig(x_dim=args.grid, y_dim=args.grid)
    W = H = args.grid
    B, T, R = args.envs, args.episode_steps, args.window
    alts = [float(a) for a in range(5, 15)]
    tanx, tany = math.tan(math.radians(cfg.angle_x) / 2), math.tan(math.radians(cfg.angle_y) / 2)
    V = args.vec
    pw = args.pw or min((W + 1) & ~1, ((2 * R + 5 + 2) // 2) * 2)
    if V == 4 and not args.pw:
        pw = min((W + 3) & ~3, (2 * R + 5 + 3 + 3 + 3) & ~3)  # widest footprint 5 + 2 R, first column rounded down, last rounded up
    ph = min(H, 2 * R + 6)
    pstride = (pw * ph + 15) & ~15
    rank_cap = 9 * T
    t_now = args.step
    acts = {t: cell_centre_actions(cfg, t, 0, B, B, alts) for t in range(t_now - T, t_now + 1)}
    phase = np.arange(B) % T

    def rect_of(fp):
        xl, xr, yu, yd, rf = fp
        r0, r1 = max(0, yu - R), min(H - 1, yd + R)
        c0, c1 = max(0, xl - R) & ~(V - 1), min(W - 1, min(W - 1, xr + R) | 1)
        if V == 4:
            c1 = min(W - 1, xr + R) | 3  # (virtual columns W .. beyond an edge that is no multiple of 4: stored zeros)
        return r0, r1, c0, c1

    tot = collections.Counter()
    col_hist = collections.Counter()   # columns of overlap between a request's unit rectangle and the stored rectangle
    row_hist = collections.Counter()
    lanes_hist = collections.Counter()
    for e in range(B):
        w = (t_now + phase[e]) % T  # steps done in this episode before the modelled one
        # stored steps of the episode: (rect, m, first column index)
        stored, k = [], 0
        for s in range(t_now - w, t_now):
            fp = footprint(W, H, cfg.resolution, tanx, tany, cfg.rf_altitude, acts[s][e])
            nx, ny = (fp[1] - fp[0]) // fp[4] + 1, (fp[3] - fp[2]) // fp[4] + 1
            stored.append((rect_of(fp), nx * ny, k))
            k += nx * ny
        fp = footprint(W, H, cfg.resolution, tanx, tany, cfg.rf_altitude, acts[t_now][e])
        xl, xr, yu, yd, rf = fp
        m = ((xr - xl) // rf + 1) * ((yd - yu) // rf + 1)
        r0n, r1n, c0n, c1n = rect_of(fp)
        hn, wn = r1n - r0n + 1, c1n - c0n + 1
        UC = 64 * V  # cells of a unit
        n_units = (hn * wn + UC - 1) // UC
        contrib = [(rc, ms, k0) for rc, ms, k0 in stored if rc[0] <= yd and rc[1] >= yu and rc[2] <= xr and rc[3] >= xl]
        n_c = sum(ms for _, ms, _ in contrib)
        tot["items"] += 1; tot["units"] += n_units; tot["records"] += n_c; tot["rank"] += k
        # gather: per contributing column, the footprint cells inside its rectangle: 4-byte loads; lines = distinct per (column, footprint row)
        for (a0, a1, b0, b1), ms, k0 in contrib:
            for j in range(ms):
                base = (k0 + j) * pstride * 4
                fr = [y for y in range(max(yu, a0), min(yd, a1) + 1)]
                fc0, fc1 = max(xl, b0), min(xr, b1)
                if not fr or fc1 < fc0:
                    continue
                tot["gather_useful"] += len(fr) * (fc1 - fc0 + 1) * 4
                for y in fr:
                    lo = base + ((y - a0) * pw + (fc0 - b0)) * 4
                    hi = base + ((y - a0) * pw + (fc1 - b0)) * 4 + 3
                    tot["gather_lines128"] += hi // 128 - lo // 128 + 1
                    tot["gather_sect64"] += hi // 64 - lo // 64 + 1
        # units
        idx = V * np.arange(64)
        LB = 4 * V  # bytes per lane request
        for u in range(n_units):
            cell = u * UC + idx
            prow, pcol = cell // wn, cell % wn
            valid = prow < hn
            rrow, rcol = r0n + np.minimum(prow, hn - 1), c0n + pcol
            urow0, urow1 = r0n + (u * UC) // wn, r0n + min(hn - 1, (u * UC + UC - 1) // wn)
            n_valid = int(valid.sum())
            # mean / variance: 8 bytes per lane and plane, planes of W floats per row
            a = ((rrow * W + rcol) * 4)[valid]
            for _ in range(2):
                tot["md_useful"] += n_valid * LB
                tot["md_lines128"] += len(np.unique(np.concatenate([a // 128, (a + LB - 1) // 128])))
                tot["md_sect64"] += len(np.unique(np.concatenate([a // 64, (a + LB - 1) // 64])))
            # new rows + planes written
            tot["write_useful"] += n_valid * LB * (m + 2)
            nact = 0
            for (a0, a1, b0, b1), ms, k0 in contrib:
                if a1 < urow0 or a0 > urow1:
                    continue
                inr = valid & (rrow >= a0) & (rrow <= a1) & (rcol >= b0) & (rcol <= b1)
                nl = int(inr.sum())
                nact += ms
                lanes_hist[nl // 8 * 8] += ms
                if nl == 0:
                    tot["rows_empty"] += ms
                    continue
                cols = int(rcol[inr].max() - rcol[inr].min()) + V
                rows = int(rrow[inr].max() - rrow[inr].min()) + 1
                col_hist[cols] += ms; row_hist[rows] += ms
                rel_r, rel_c = (rrow - a0)[inr], (rcol - b0)[inr]
                for j in range(ms):
                    base = (k0 + j) * pstride * 4
                    if args.layout == "row":
                        ad = base + (rel_r * pw + rel_c) * 4
                    elif args.layout == "half":  # two half-width patches (pw / 2 columns each), the second behind the first
                        hw = pw // 2
                        ad = base + np.where(rel_c < hw, (rel_r * hw + rel_c) * 4, ph * hw * 4 + (rel_r * hw + (rel_c - hw)) * 4)
                    elif args.layout == "tile4x8":  # tiles of 4 rows x 8 columns = 128 bytes, tiles row-major over ceil(pw / 8) tile columns
                        tc = (pw + 7) // 8
                        ad = base + (((rel_r // 4) * tc + rel_c // 8) * 32 + (rel_r % 4) * 8 + rel_c % 8) * 4
                    else:
                        raise SystemExit("unknown layout")
                    tot["row_useful"] += nl * LB
                    tot["row_lines128"] += len(np.unique(np.concatenate([ad // 128, (ad + LB - 1) // 128])))
                    tot["row_sect64"] += len(np.unique(np.concatenate([ad // 64, (ad + LB - 1) // 64])))
            tot["row_requests"] += nact
            tot["alg_floats"] += (nact + m + 4) * n_valid * V
            tot["lane_slots"] += nact * 64
    n = tot["items"]
    MB = 1e-6
    print(f"{n} items of step {t_now}: layout {args.layout}, pw {pw}, ph {ph}, pstride {pstride} floats; mean rank before {tot['rank'] / n:.1f}, records {tot['records'] / n:.1f} per item, "
          f"{tot['units'] / n:.2f} units per item, row requests {tot['row_requests'] / tot['units']:.1f} per unit ({tot['rows_empty'] / max(tot['row_requests'], 1):.1%} of them with no lane inside the rectangle)")
    print(f"SURVEY count (algorithmic) {tot['alg_floats'] * 4 * MB:7.1f} MB per launch; row requests (wave instructions) {tot['row_requests'] / n:.1f} per item, "
          f"lanes inside a stored rectangle {tot['row_useful'] / (4 * V) / max(tot['lane_slots'], 1):.1%} of the request lanes")
    for name, key in (("row stream", "row"), ("gather", "gather"), ("mean / variance reads", "md")):
        u, l, s = tot[key + "_useful"] * MB, tot[key + "_lines128"] * 128 * MB, tot[key + "_sect64"] * 64 * MB
        print(f"{name:24s} useful {u:7.1f} MB   128-byte lines {l:7.1f} MB ({l / u:4.2f} x)   64-byte sectors {s:7.1f} MB ({s / u:4.2f} x)")
    print(f"{'writes (rows + planes)':24s} useful {tot['write_useful'] * MB:7.1f} MB")
    tc = sum(col_hist.values())
    print("columns of overlap per row request (lanes inside the stored rectangle span this many columns): share of requests")
    acc = 0
    for c in sorted(col_hist):
        acc += col_hist[c]
        print(f"   {c:3d} columns: {col_hist[c] / tc:6.1%}   cumulative {acc / tc:6.1%}")
    print(f"   mean {sum(c * v for c, v in col_hist.items()) / tc:.1f} of {pw} columns; rows per request: mean {sum(c * v for c, v in row_hist.items()) / tc:.2f}")
    print("lanes inside the rectangle per request (of 64): " + "  ".join(f"{k}-{k + 7}: {v / sum(lanes_hist.values()):.1%}" for k, v in sorted(lanes_hist.items())))


if __name__ == "__main__":
    main()
