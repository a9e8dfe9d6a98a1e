#!/usr/bin/env python3
"""
bench.py -- batched IPP env-steps/s on MI355X (BASELINE.json metric), one process per GPU.

Workload (config.workload): BASELINE.json configs[1] -- 4096 parallel envs per GPU, 50x50 grid, 10 altitude
levels 5..14 m, example.yaml sensor / prior / UAV parameters, Gaussian-random-field ground truths, adaptive
masked reward with flight-time cost.  One bench "step" = one fused env step of every env of
the batch (predict + observe + mean/covariance update + reward) including the episode resets that fall on
that step (episodes are 40 steps long and staggered so every step sees the stationary mix of factor ranks).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = envs of all ranks x K / max-over-ranks wall time of the K timed steps
(inputs resident in HBM).  `roofline` is measured live with HIP events around the streaming kernel;
`cpu_baseline` times the plain-C fp64 oracle (oracle/ipp_oracle.c, a port of the reference's NumPy path) on the
host cores on a bounded sample of the same workload.  Synthetic data, device Philox noise.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
ALTITUDES = [float(a) for a in range(5, 15)]  # 10 levels, min 5, max 14, spacing 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (weak scaling)")
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--state", choices=["factor", "dense"], default="factor")
    ap.add_argument("--episode-steps", type=int, default=40)
    ap.add_argument("--tile-threads", type=int, default=0)
    ap.add_argument("--window-rows", type=int, default=-1,
                    help="factor state: keep new columns of U within R grid rows of the footprint; -1 = the smallest R for which "
                         "the dropped prior covariances stay below 1e-6 (ipp_min_window_rows: 10 for the example prior, which the "
                         "bench never rescales; parity-tested at 1e-5); 0 = exact full columns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=0)
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--predict-only", action="store_true", help="time the predict-only rate (tree-search call)")
    ap.add_argument("--fused-resets", action="store_true",
                    help="A/B: scheduled episode resets inside the step launch (ipp_step_autoreset) instead of their own launch")
    ap.add_argument("--print-args", action="store_true", help="print the workload key used to match PMC summaries, then exit")
    return ap.parse_args()


def baseline_config_name(args):
    """Which BASELINE.json config the command line is."""
    if args.grid == 50 and args.envs == 4096 and args.episode_steps == 40:
        return "BASELINE configs[1]"
    if args.grid == 100 and args.envs == 32768 and args.episode_steps == 16:
        return "BASELINE configs[2]"
    if args.grid == 50 and args.envs == 32768 and args.episode_steps == 40:
        return "BASELINE configs[3] (per-GPU share)"
    return "custom config"


def workload_key(args):
    return {"envs": args.envs, "grid": args.grid, "state": args.state, "window_rows": args.window_rows,
            "tile_threads": args.tile_threads, "episode_steps": args.episode_steps, "predict_only": bool(args.predict_only)}


def cpu_baseline(cfg, args):
    """Plain-C fp64 dense port of the reference path on the host cores, bounded sample of the same workload."""
    from oracle import c_oracle, ipp_oracle as orc
    from ipp_rl_amd.vec_env import cell_centre_actions

    ocfg = orc.OracleConfig(x_dim=cfg.x_dim, y_dim=cfg.y_dim)
    cores = os.cpu_count() or 1
    threads = min(cores, c_oracle.max_threads())
    B = args.cpu_envs or int(min(256, max(8, 2 * threads)))
    T = args.cpu_steps
    n = ocfg.n_cells
    P0 = c_oracle.matern_prior(ocfg)
    P = np.ascontiguousarray(np.broadcast_to(P0, (B, n, n)))
    mean = 0.5 * np.ones((B, n))
    rs = np.random.RandomState(7)
    h = orc.grf_kernel(cfg.y_dim, cfg.x_dim, 5.0)
    gts = np.stack([c_oracle.grf_from_kernel(rs.normal(size=(cfg.y_dim, cfg.x_dim)), h).ravel() for _ in range(min(B, 4))])
    gts = np.ascontiguousarray(np.resize(gts, (B, n)))
    acts = np.stack([cell_centre_actions(cfg, t, 0, B, B, ALTITUDES) for t in range(T)])
    eps = np.zeros((T, B, c_oracle.OC_MAX_M))
    eps[:, :, :9] = rs.normal(size=(T, B, 9))
    t0 = time.perf_counter()
    c_oracle.run_batch(ocfg, P, mean, gts, acts, np.array([2.0, 2.0, 14.0]), eps=eps, threads=threads)
    dt = time.perf_counter() - t0
    # single-thread rate ("(i) 1 thread; (ii) all cores") on a few env-steps of the same workload
    B1, T1 = 2, min(T, 10)
    P1 = np.ascontiguousarray(np.broadcast_to(P0, (B1, n, n)))
    t1 = time.perf_counter()
    c_oracle.run_batch(ocfg, P1, 0.5 * np.ones((B1, n)), np.ascontiguousarray(gts[:B1]), np.ascontiguousarray(acts[:T1, :B1]),
                       np.array([2.0, 2.0, 14.0]), eps=np.ascontiguousarray(eps[:T1, :B1]), threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": B * T / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
        "value_1_thread": B1 * T1 / dt1,
        "sample": f"{B} envs x {T} steps, {cfg.x_dim}x{cfg.y_dim} grid, dense fp64 state (reference representation), "
                  f"{T}-step episodes, OpenMP over envs ({threads} threads), {dt:.1f} s; 1 thread: {B1} envs x {T1} steps, {dt1:.1f} s",
    }


def pmc_traffic(kernel_name, args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS command line
    (tools/pmc_run.sh -> tools/pmc_summary.py -> profiles/*_pmc_summary.json): FETCH_SIZE x 2 (gfx950 correction,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KiB.  None when no matching summary is committed:
    counters cannot be collected from inside the timed process."""
    import glob

    want = workload_key(args)
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_summary*.json"))):
        try:
            with open(path) as fh:
                summary = json.load(fh)
        except (OSError, ValueError):
            continue
        if summary.get("_bench_args") != want:
            continue
        for name, counters in summary.items():
            if name.startswith("ipp::" + kernel_name + "<") and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
                return (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
    return None


def main():
    args = parse()
    if args.print_args:
        print(json.dumps(workload_key(args)))
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC (RCCL)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"

    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

    cfg = EngineConfig(x_dim=args.grid, y_dim=args.grid)
    B, T = args.envs, args.episode_steps
    total_envs = B * world
    lo, hi = rank * B, (rank + 1) * B  # contiguous env-id range per GPU, no exchange between shards
    env = VecIPPEnv(cfg, B, state=args.state, episode_steps=T, device=device, seed=1234, env_id_offset=lo,
                    stagger=True, tile_threads=args.tile_threads, window_rows=args.window_rows,
                    fused_reset=args.fused_resets)
    eng = env.engine
    n_total = T + args.warmup + 2 * args.steps
    # synthetic inputs resident in HBM before the timed region
    actions = torch.stack([
        torch.as_tensor(cell_centre_actions(cfg, t, lo, hi, total_envs, ALTITUDES), dtype=torch.float64)
        for t in range(n_total)
    ]).to(device)
    env.reset()
    t_idx = 0
    step_kw = {}
    if args.predict_only:
        # tree-search call: reward only, state untouched; pre-roll below still builds the stationary state
        pass
    for _ in range(T):  # pre-roll: reach the stationary mix of episode phases (untimed setup, not warmup)
        env.step(actions[t_idx]); t_idx += 1

    def run_steps(k):
        nonlocal t_idx
        for _ in range(k):
            if args.predict_only:
                eng.step(actions[t_idx], env.prev, predict_only=True, cov_only=True, reward_out=env.reward,
                         status_out=env.status)
            else:
                env.step(actions[t_idx])
            t_idx += 1

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())
    bad = int((env.status != 0).sum().item())
    bad_rewards = int((~torch.isfinite(env.reward)).sum().item())

    # ---- roofline leg: same steps again with HIP events around the streaming kernel (rank 0 reports)
    eng.profile(True)
    eng.streamed_bytes(reset=True)
    rank_sum = torch.zeros((), dtype=torch.float64, device=device)
    ranks_buf = torch.empty(B, dtype=torch.int32, device=device)
    for _ in range(args.steps):
        if args.predict_only:
            run_steps(1)
            rank_sum += eng.ranks(ranks_buf).double().sum()  # rows streamed (nothing appended)
        else:
            # rows streamed + columns appended by this step = ranks right after the step kernel and before
            # the scheduled resets: env.step() snapshots them through this hook
            env.step(actions[t_idx], after_step_hook=lambda: rank_sum.add_(eng.ranks(ranks_buf).double().sum()))
            t_idx += 1
    torch.cuda.synchronize()
    counted_bytes = eng.streamed_bytes(reset=True) / args.steps  # device counter: rows x valid cells actually streamed
    gain_ms, gain_n = eng.profile_read(0)
    down_ms, down_n = eng.profile_read(1)
    prep_ms, prep_n = eng.profile_read(2)
    eng.profile(False)
    N = cfg.n_cells
    mean_rank_after = float(rank_sum.item()) / (args.steps * B)
    if args.state == "factor":
        # 4N(r + m) + 16N bytes per committed step; predict-only reads 4N r + 8N (mean, diag)
        per_step = 4.0 * N * mean_rank_after + (8.0 * N if args.predict_only else 16.0 * N)
        kernel_ms, kernel_name = gain_ms, "k_gain"
        if int(eng.info.window_rows) > 0:  # mirrors the kernel selection in csrc/ipp_engine.hip launch_chunk()
            tt = int(eng.info.tile_threads)
            fused = tt == 256 and os.environ.get("IPP_FUSED", "1") != "0"
            kernel_name = "k_step_factor" if fused else ("k_gain_wave" if tt == 64 else "k_gain_factor")
    else:
        per_step = (4.0 * N * 25 + 8.0 * N) if args.predict_only else (8.0 * N * N + 16.0 * N)
        kernel_ms, kernel_name = (gain_ms, "k_gain") if args.predict_only else (down_ms, "k_downdate")
    formula_bytes = per_step * B  # SURVEY 8(d) formula with full columns
    # factor state: the device counter holds what the launch really streamed (windowed columns, SURVEY 8(d): "N must be
    # replaced by the window size actually streamed"); dense state: the formula is exact
    bytes_per_launch = counted_bytes if args.state == "factor" else formula_bytes
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "env-steps/s (batched) on 50x50 grid" if args.grid == 50 else f"env-steps/s (batched) on {args.grid}x{args.grid} grid",
            "value": total_envs * args.steps / elapsed_max,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{baseline_config_name(args)}: {B} parallel envs per GPU, {args.grid}x{args.grid} grid, 10 altitude "
                            f"levels 5-14 m, example.yaml sensor/prior/UAV, GRF ground truth, adaptive reward with "
                            f"flight-time cost; {'predict-only (reward) calls' if args.predict_only else 'full fused env step (predict + observe + update)'}",
                "envs_per_gpu": B, "grid": f"{args.grid}x{args.grid}", "state_repr": args.state,
                "episode_steps": T, "episode_phase": "staggered (stationary rank mix)",
                "mean_rank_after_step": mean_rank_after, "tile_threads": int(eng.info.tile_threads), "window_rows": int(eng.info.window_rows),
                "items_with_nonzero_status": bad, "non_finite_rewards": bad_rewards, "rng": "device Philox4x32-10",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kernel_name, args),
                "kernel": kernel_name, "kernel_ms_avg": kernel_ms, "launches": down_n if kernel_name == "k_downdate" else gain_n,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "full_column_formula_bytes_per_launch": formula_bytes,
                "other_kernels_ms_avg": {"k_prepare": prep_ms, "k_gain": gain_ms, "k_downdate": down_ms},
            },
        }
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, args)
            except Exception as exc:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
