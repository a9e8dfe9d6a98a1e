#!/usr/bin/env python3
"""
bench.py -- batched IPP env-steps/s on MI355X (BASELINE.json metric), one process per GPU.

Workload (config.workload): BASELINE.json configs[1] -- 4096 parallel envs per GPU, 50x50 grid, 10 altitude
levels 5..14 m, example.yaml sensor / prior / UAV parameters, Gaussian-random-field ground truths, adaptive
masked reward with flight-time cost.  One bench "step" = one fused env step of every env of
the batch (predict + observe + mean/covariance update + reward) including the episode resets that fall on
that step (episodes are 40 steps long and staggered so every step sees the stationary mix of factor ranks).

  python bench.py [--gpus N --steps K --warmup W]
      N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (before it touches the
      GPU) as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` and relays
      rank 0's JSON line
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W          (the driver's form: ranks read RANK / LOCAL_RANK / WORLD_SIZE)

Rank 0 prints ONE JSON line.  `value` = envs of all ranks x K / max-over-ranks wall time of the K timed steps
(inputs resident in HBM).  `roofline` is measured live with HIP events attached to the streaming kernel's
dispatches; `cpu_baseline` times the plain-C fp64 oracle (oracle/ipp_oracle.c, a port of the reference's NumPy path)
on the host cores on a bounded sample of the same workload.  At N = 1 the line also carries `extra`: the same
measurement for the other BASELINE.json configs that fit one GPU (configs[2], the per-GPU share of configs[3], a
configs[4] tree-search wave) and for the headline's neighbours (window 12, predict-only).  Synthetic data, device
Philox noise keyed on the global env id.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
ALTITUDES = [float(a) for a in range(5, 15)]  # 10 levels, min 5, max 14, spacing 1


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (a region of 200 steps is 20 ms: past the clock ramp that spread 80-step regions by 17 %)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (weak scaling)")
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong scaling: this many envs in total, split into contiguous env-id ranges over the ranks "
                         "(BASELINE configs[3]: --envs-total 262144 --gpus 8)")
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--state", choices=["factor", "dense"], default="factor")
    ap.add_argument("--episode-steps", type=int, default=40)
    ap.add_argument("--tile-threads", type=int, default=0)
    ap.add_argument("--window-rows", type=int, default=-1,
                    help="factor state: keep new columns of U within R grid rows of the footprint; -1 = the smallest R for which "
                         "the dropped prior covariances stay below 1e-6 (ipp_min_window_rows: 10 for the example prior, which the "
                         "bench never rescales; 12 with --shuffle-prior; parity-tested at 1e-5); 0 = exact full columns")
    ap.add_argument("--shuffle-prior", action="store_true",
                    help="episodes draw (sigma^2, l) in [0.8, 1.2] x nominal like the reference's self-play "
                         "(planning/mcts_zero/episode_generators.py:53): the window has to hold for 1.2 l -> 12 rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` records (other configs) of the N = 1 line")
    ap.add_argument("--cpu-envs", type=int, default=0)
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--predict-only", action="store_true", help="time the predict-only rate (tree-search call)")
    ap.add_argument("--fused-resets", action="store_true", default=True,
                    help="scheduled episode resets inside the step launch (ipp_step_autoreset; VecIPPEnv's default)")
    ap.add_argument("--no-fused-resets", dest="fused_resets", action="store_false",
                    help="A/B: the scheduled resets as their own launch after every step")
    ap.add_argument("--parts", type=int, default=2,
                    help="schedule of a step: the batch as this many fixed groups of envs, one launch and one stream per group "
                         "(VecIPPEnv.step_async: a group's step t + 1 is ordered behind its OWN step t only -- envs are independent --, so the "
                         "next launch of one group fills the slots that the slowest items of the other still hold); 1 = one launch per step on "
                         "one stream (rounds 1-3).  Measured on MI355X at configs[1] (profiles/r04_experiments.txt 10-13): two groups 55-56 M "
                         "env-steps/s against 45 M for one launch, provided the two part streams and the ground-truth staging stream sit on "
                         "three different hardware queues (VecIPPEnv probes for that; on a shared queue 43-47 M); three groups equal two.  "
                         "With parts > 1 the one-launch rate of the same env is measured as well (config.sync_schedule) and the roofline leg "
                         "times both forms")
    ap.add_argument("--step-priority", type=int, default=0,
                    help="A/B: run the steps on a stream of this priority (-1: above the side stream that generates the next episodes' "
                         "ground truths, whose workgroups then only take the slots the step launches leave free)")
    ap.add_argument("--regions", type=int, default=7,
                    help="timed regions of --steps steps each (barrier + sync around every one); `value` is the median region")
    ap.add_argument("--print-args", action="store_true", help="print the workload key used to match PMC summaries, then exit")
    return ap


def parse(argv=None):
    return build_parser().parse_args(argv)


# --------------------------------------------------------------------------------------------- multi-GPU plumbing
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(args, argv, port):
    """The command line that starts args.gpus ranks of this script on one node (what the driver runs for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks as child processes and exit with their
    code.  Nothing in this process has touched the GPU (torch is not even imported yet); the ranks are fresh
    processes, never an exec of a GPU-initialised one."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(launch_command(args, argv, free_port()), env=env)
    sys.exit(proc.returncode)


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD display / processing-accelerator PCI function, in PCI address order (= the HIP device order on one
    node unless HIP_VISIBLE_DEVICES reorders it); -1 where the platform does not say.  Reads sysfs only: no GPU call."""
    import glob

    nodes = []
    for dev in sorted(glob.glob(os.path.join(sysfs, "bus/pci/devices/*"))):
        if not dev.endswith(".0"):
            continue  # (one entry per device: function 0)
        try:
            with open(os.path.join(dev, "vendor")) as fh:
                if fh.read().strip() != "0x1002":
                    continue
            with open(os.path.join(dev, "class")) as fh:
                cls = fh.read().strip()
            if not (cls.startswith("0x03") or cls.startswith("0x12")):
                continue
            with open(os.path.join(dev, "numa_node")) as fh:
                nodes.append(int(fh.read().strip()))
        except (OSError, ValueError):
            continue
    return nodes


def plan_rank_cores(local_rank, local_world, allowed, gpu_nodes=None, node_cpus=None):
    """The cores rank `local_rank` of `local_world` ranks on this node pins itself to: DISJOINT sets; when the platform tells which
    NUMA node each GPU hangs on (gpu_nodes[i], node_cpus[node]), a rank takes its share of ITS GPU's node -- the ranks of one node
    split that node's allowed cores evenly -- else an even contiguous split of the allowed cores.  Returns (cores, how)."""
    allowed = sorted(allowed)
    if local_world <= 1 or len(allowed) < local_world:
        return allowed, "all allowed cores (one rank, or fewer cores than ranks)"
    if gpu_nodes and node_cpus and len(gpu_nodes) >= local_world and all(n >= 0 for n in gpu_nodes[:local_world]):
        mine = gpu_nodes[local_rank]
        peers = [r for r in range(local_world) if gpu_nodes[r] == mine]
        cpus = [c for c in node_cpus.get(mine, []) if c in set(allowed)]
        if len(cpus) >= len(peers):
            k, per = peers.index(local_rank), len(cpus) // len(peers)
            return cpus[k * per:(k + 1) * per], f"NUMA node {mine} of GPU {local_rank}: share {k} of {len(peers)}"
    per = len(allowed) // local_world
    return allowed[local_rank * per:(local_rank + 1) * per], "even contiguous split (no NUMA information)"


def pin_rank(local_rank, local_world):
    """Pins this process (before anything touches the GPU or starts a thread pool) to its cores; the report goes on the line.
    At 8 ranks the value is the MAX over ranks of a host-issued loop: a rank whose Python thread is descheduled or migrated across
    sockets is the job's number.  IPP_BENCH_PIN=0: leave the affinity alone."""
    if local_world <= 1 or os.environ.get("IPP_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    allowed = os.sched_getaffinity(0)
    node_cpus = {}
    try:
        import glob

        for nd in glob.glob("/sys/devices/system/node/node[0-9]*"):
            with open(os.path.join(nd, "cpulist")) as fh:
                node_cpus[int(os.path.basename(nd)[4:])] = _parse_cpulist(fh.read())
    except (OSError, ValueError):
        node_cpus = {}
    cores, how = plan_rank_cores(local_rank, local_world, allowed, gpu_numa_nodes(), node_cpus)
    try:
        os.sched_setaffinity(0, cores)
    except OSError as exc:
        return {"pinned": False, "cores": len(allowed), "error": repr(exc)}
    return {"pinned": True, "cores": len(cores), "first_core": min(cores), "how": how}


def shard_plan(args, rank, world):
    """Contiguous env-id range of `rank`: weak scaling = args.envs per rank, strong scaling
    (--envs-total) = the total split evenly.  Returns (lo, hi, total_envs, scaling)."""
    from ipp_rl_amd.vec_env import shard_range

    if args.envs_total > 0:
        lo, hi = shard_range(args.envs_total, rank, world)
        return lo, hi, args.envs_total, "strong"
    return rank * args.envs, (rank + 1) * args.envs, args.envs * world, "weak"


class Ranks:
    """torch.distributed used for exactly three things: the barrier around the timed region, the MAX over ranks of its
    wall time, and gathering the per-rank times for the report.  No data-path collective exists (envs are
    independent).  backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU test of this class."""

    def __init__(self, backend="nccl", device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device(f"cuda:{self.local_rank}")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def all_times(self, elapsed):
        """([elapsed of every rank], their max)"""
        if self.dist is None:
            return [elapsed], elapsed
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device=self.device or "cpu")
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        tmax = t.clone()
        self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
        return [float(x.item()) for x in out], float(tmax.item())

    def gather(self, x: float):
        """[x of every rank] (one small all_gather, outside every timed region)."""
        if self.dist is None:
            return [float(x)]
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device or "cpu")
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(v.item()) for v in out]

    def all_agree(self, flag: bool) -> bool:
        """True iff `flag` holds on EVERY rank (a choice that changes how many collectives a rank enters must be the same everywhere)."""
        if self.dist is None:
            return bool(flag)
        import torch

        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def timed_region(run_steps, steps, sync, ranks, info=None):
    """Time EXACTLY `steps` steps bracketed by barrier + device sync on both sides; returns (per-rank seconds,
    max over ranks).  info (optional list): receives this rank's (host time to ISSUE the steps -- before the closing sync --,
    time until its OWN device had finished them -- before the closing barrier).  The per-rank seconds include the wait at that
    barrier, so they are nearly equal on every rank: which rank the job waited for is in the second figure."""
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps(steps)
    t_issue = time.perf_counter() - t0
    sync()
    t_own = time.perf_counter() - t0
    ranks.barrier()
    sync()
    if info is not None:
        info.append((t_issue, t_own))
    return ranks.all_times(time.perf_counter() - t0)


def aggregate_rate(total_units_per_step, steps, elapsed_max):
    """Whole-job throughput: units of ALL ranks / the slowest rank's time."""
    return total_units_per_step * steps / elapsed_max


# --------------------------------------------------------------------------------------------- workload description
def baseline_config_name(grid, envs, episode_steps, total=None):
    """Which BASELINE.json config the command line is."""
    if grid == 50 and envs == 4096 and episode_steps == 40:
        return "BASELINE configs[1]"
    if grid == 100 and envs == 32768 and episode_steps == 16:
        return "BASELINE configs[2]"
    if grid == 50 and envs == 32768 and episode_steps == 40:
        return "BASELINE configs[3]" + (" (per-GPU share)" if not total or total == envs else "")
    return "custom config"


def workload_key(args):
    return {"envs": args.envs, "grid": args.grid, "state": args.state, "window_rows": args.window_rows,
            "tile_threads": args.tile_threads, "episode_steps": args.episode_steps, "predict_only": bool(args.predict_only),
            "shuffle_prior": bool(args.shuffle_prior), "parts": args.parts}


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


def pmc_traffic(kernel_name, key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS command line
    (tools/pmc_run.sh -> tools/pmc_summary.py -> profiles/*_pmc_summary*.json): FETCH_SIZE x 2 (gfx950 correction,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KiB.  Returns (bytes, source file) -- (None, None) when no
    matching summary is committed: counters cannot be collected from inside the timed process, so this number is
    REPLAYED from the named file, not measured in this run."""
    import glob

    # the counter passes run the batch as ONE launch per step (--parts 1: per-dispatch counters of overlapping launches are not
    # separable, and the groups' launches of a step move the same bytes as the single launch)
    key = dict(key, parts=1)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.json")), reverse=True):
        try:
            with open(path) as fh:
                summary = json.load(fh)
        except (OSError, ValueError):
            continue
        if summary.get("_bench_args") != key:
            continue
        for name, counters in summary.items():
            if name.startswith("ipp::" + kernel_name + "<") and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
                return (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


# --------------------------------------------------------------------------------------------- the measurement
def run_env_workload(torch, ranks, device, *, grid, envs_local, env_lo, total_envs, episode_steps, state="factor",
                     window_rows=-1, shuffle_prior=False, tile_threads=0, predict_only=False, fused_resets=True,
                     steps=80, warmup=8, timed=True, regions=1, parts=1):
    """One workload: build the batched env, pre-roll to the stationary rank mix, W warm-up steps, the timed region
    (all ranks), then the roofline leg (same steps again with HIP events on the streaming kernel's dispatches).
    Returns a dict of plain numbers."""
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    B, T = envs_local, episode_steps
    env = VecIPPEnv(cfg, B, state=state, episode_steps=T, device=device, seed=1234, env_id_offset=env_lo,
                    stagger=True, tile_threads=tile_threads, window_rows=window_rows, fused_reset=fused_resets,
                    shuffle_prior_cov=shuffle_prior, parts=parts)
    eng = env.engine
    # (a runtime with fewer hardware queues than groups -- GPU_MAX_HW_QUEUES -- makes the groups' launches take turns: one launch per step then)
    # the schedule is the job's, not a rank's: a rank whose streams had to share a queue takes every rank to one launch per step
    # (the one-launch legs add timed regions = barriers: a per-rank choice would leave ranks in different collectives)
    use_parts = ranks.all_agree(env.parts > 1 and bool(env._fused_reset) and env.part_queues_distinct)
    n_total = T + warmup + (2 * regions + 3) * steps
    # synthetic inputs resident in HBM before the timed region
    actions = torch.stack([
        torch.as_tensor(cell_centre_actions(cfg, t, env_lo, env_lo + B, total_envs, ALTITUDES), dtype=torch.float64)
        for t in range(n_total)
    ]).to(device)
    env.reset()
    t_idx = 0

    from ipp_rl_amd import _ffi

    predict_flags = _ffi.IPP_COV_ONLY | _ffi.IPP_PREDICT_ONLY | _ffi.IPP_ADAPTIVE | _ffi.IPP_USE_FLIGHT_TIME

    def run_steps(k):
        nonlocal t_idx
        for _ in range(k):
            if predict_only and use_parts:
                # rewards of candidate actions, no state write: the two groups' launches alternate on their queues (ipp_step_parts)
                eng.set_item_order(env._orders_parts[env.t % T])
                eng.step_parts(actions[t_idx], env.prev, None, predict_flags, env.reward, env.status, env._part_begin, env._part_streams)
            elif predict_only:
                eng.step(actions[t_idx], env.prev, predict_only=True, cov_only=True, reward_out=env.reward,
                         status_out=env.status)
            elif use_parts:
                # (the actions were written and synchronised before the timed region: nothing to order against)
                env.step_async(actions[t_idx], inputs_ready=True)
            else:
                env.step(actions[t_idx])
            t_idx += 1

    def run_steps_sync(k):  # one launch per step on one stream (the schedule of rounds 1-3), same env, same kernels
        nonlocal t_idx
        for _ in range(k):
            if predict_only:
                eng.set_item_order(env._orders[env.t % T])
                eng.step(actions[t_idx], env.prev, predict_only=True, cov_only=True, reward_out=env.reward, status_out=env.status)
            else:
                env.step(actions[t_idx], after_step_hook=_no_hook)
            t_idx += 1

    def median_region(regs):
        order_r = sorted(range(len(regs)), key=lambda i: regs[i][1])
        return regs[order_r[len(order_r) // 2]]

    # pre-roll: reach the stationary mix of episode phases (untimed setup, not warmup) -- issued the way the timed regions issue their
    # steps (whole runs of `steps` steps without a join), so that whatever the runtime sets up the first time that many launches are in
    # flight happens here: the first process on a fresh box spent 35 ms of HOST time inside its first timed region otherwise
    # (1.75 ms per step in a 20-step region; profiles/r06_experiments.txt 15)
    done = 0
    while done < T:
        k = min(max(steps, 1), T - done)
        if predict_only:  # (the states are built by COMMITTED steps; the predict-only calls of the timed regions write nothing)
            for _ in range(k):
                env.step(actions[t_idx]); t_idx += 1
        else:
            run_steps(k)
        done += k
    run_steps(warmup)
    # `regions` timed regions of EXACTLY `steps` steps each, every one bracketed by barrier + device sync on both sides; the
    # reported region is the MEDIAN one (by the max-over-ranks time), the spread goes into the record
    def sync_all():  # join the part streams into the caller's, then the device (the env learns that nothing of it is in flight any more)
        env.wait()
        torch.cuda.synchronize()

    issue_s = []  # host time to issue a region's steps (the launches queue up behind the device: issue time < region time = device-bound)
    timed_regions = [timed_region(run_steps, steps, sync_all, ranks, issue_s) for _ in range(max(1, regions))]
    per_rank, elapsed_max = median_region(timed_regions)
    # every rank's host time to issue a step (median region): at N ranks the slowest HOST loop can be what the MAX over ranks reports
    med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
    issue_per_rank = ranks.gather(1e3 * med([i for i, _ in issue_s]) / steps)
    own_per_rank = ranks.gather(1e3 * med([o for _, o in issue_s]) / steps)  # (without the wait for the other ranks at the closing barrier)
    sync_regions = None
    if use_parts:
        env.wait()
        sync_regions = [timed_region(run_steps_sync, steps, torch.cuda.synchronize, ranks) for _ in range(max(1, regions))]
    bad = int((env.status != 0).sum().item())
    bad_rewards = int((~torch.isfinite(env.reward)).sum().item())

    # ---- roofline leg: same steps again with HIP events around the streaming kernel (every rank runs it, rank 0 reports)
    eng.profile(True)
    eng.streamed_bytes(reset=True)
    rank_sum = torch.zeros((), dtype=torch.float64, device=device)
    ranks_buf = torch.empty(B, dtype=torch.int32, device=device)
    for _ in range(steps):
        if predict_only:
            run_steps_sync(1)
            rank_sum += eng.ranks(ranks_buf).double().sum()  # rows streamed (nothing appended)
        else:
            # rows streamed + columns appended by this step = ranks right after the step kernel and before
            # the scheduled resets: env.step() snapshots them through this hook
            env.step(actions[t_idx], after_step_hook=lambda: rank_sum.add_(eng.ranks(ranks_buf).double().sum()))
            t_idx += 1
    torch.cuda.synchronize()
    counted, mask_reread = eng.streamed_bytes_detail(reset=True)  # device counters: floats actually streamed x 4
    needed = eng.streamed_bytes_needed() / steps  # ... with every stored row taken on the cells inside its own rectangle only
    counted /= steps
    mask_reread /= steps
    gain_ms, gain_n = eng.profile_read(0)
    down_ms, down_n = eng.profile_read(1)
    prep_ms, prep_n = eng.profile_read(2)
    eng.profile_read_busy(0)
    # ---- the same leg on the partitioned schedule: the launches of different streams overlap, so the kernel time of a step is
    # the time during which at least one of them runs (union of the dispatches' HIP-event intervals), not the sum of the durations
    busy_ms_per_step, part_launches, part_ms_avg = None, 0, None
    if use_parts:
        run_steps(steps)
        torch.cuda.synchronize()
        part_ms_avg, _ = eng.profile_read(0)
        busy, part_launches = eng.profile_read_busy(0)
        busy_ms_per_step = busy / steps
        eng.streamed_bytes(reset=True)
    eng.profile(False)
    N = cfg.n_cells
    mean_rank_after = float(rank_sum.item()) / (steps * B)
    if state == "factor":
        # 4N(r + m) + 16N bytes per committed step; predict-only reads 4N r + 8N (mean, diag)
        per_step = 4.0 * N * mean_rank_after + (8.0 * N if predict_only else 16.0 * N)
        kernel_ms, kernel_name = gain_ms, "k_gain"
        if int(eng.info.window_rows) > 0:  # mirrors the kernel selection in csrc/ipp_engine.hip launch_chunk()
            tt = int(eng.info.tile_threads)
            fused = tt == 256 and os.environ.get("IPP_FUSED", "1") != "0"
            kernel_name = "k_step_factor" if fused else ("k_gain_wave" if tt == 64 else "k_gain_factor")
            if int(eng.info.patch_layout):
                kernel_name = "k_step_patch"
    else:
        per_step = (4.0 * N * 25 + 8.0 * N) if predict_only else (8.0 * N * N + 16.0 * N)
        kernel_ms, kernel_name = (gain_ms, "k_gain") if predict_only else (down_ms, "k_downdate")
    formula_bytes = per_step * B  # SURVEY 8(d) formula with full columns
    # factor state: the device counter holds what the launch really streamed (windowed columns, SURVEY 8(d): "N must be
    # replaced by the window size actually streamed"), r + m + 4 floats per touched cell; dense state: the formula is exact
    bytes_per_launch = counted if state == "factor" else formula_bytes
    single_launch_ms = kernel_ms
    if busy_ms_per_step:
        kernel_ms = busy_ms_per_step  # per step: every part's launch, overlapped as scheduled
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    rec = {
        "parts": env.parts if use_parts else 1, "single_launch_ms": single_launch_ms, "part_launches": part_launches,
        "part_launch_ms_avg": part_ms_avg,
        "sync_region_elapsed_max_s": [r[1] for r in sync_regions] if sync_regions else None,
        "sync_elapsed_max_s": median_region(sync_regions)[1] if sync_regions else None,
        "grid": grid, "envs_local": B, "episode_steps": T, "state": state, "predict_only": bool(predict_only),
        "window_rows": int(eng.info.window_rows), "tile_threads": int(eng.info.tile_threads),
        "per_rank_s": per_rank, "elapsed_max_s": elapsed_max, "steps": steps, "warmup": warmup,
        "region_elapsed_max_s": [r[1] for r in timed_regions], "region_issue_s": [i for i, _ in issue_s], "issue_ms_per_rank": issue_per_rank, "own_ms_per_rank": own_per_rank,
        "queues": getattr(env, "queue_report", None),
        "split_min_items": int(getattr(eng.info, "patch_split_min_items", 0)),
        "mean_rank_after_step": mean_rank_after, "bad_status": bad, "bad_rewards": bad_rewards,
        "kernel": kernel_name, "kernel_ms": kernel_ms, "launches": down_n if kernel_name == "k_downdate" else gain_n,
        "bytes_per_launch": bytes_per_launch, "mask_reread_bytes_per_launch": mask_reread if state == "factor" else 0.0,
        "necessary_bytes_per_launch": needed if (state == "factor" and needed > 0) else None,
        # (patch kernel: three waves per item, two for launches of at least ipp_info.patch_two_wave_min_items items)
        "workgroup_threads": (64 * (2 if 0 < int(getattr(eng.info, "patch_two_wave_min_items", 0)) <= B // (env.parts if use_parts else 1)
                                    else int(eng.info.patch_waves))) if (state == "factor" and int(eng.info.patch_layout)) else int(eng.info.tile_threads),
        "formula_bytes_per_launch": formula_bytes, "achieved_gbs": achieved,
        "other_kernels_ms_avg": {"k_prepare": prep_ms, "k_gain": gain_ms, "k_downdate": down_ms},
        "arena_gb": float(eng.info.arena_bytes) / 1e9,
        "arena_kind": getattr(eng, "arena_kind", None), "arena_chunk_mib": (getattr(eng.arena, "chunk_bytes", 0) >> 20) or None,
    }
    env.close()
    del env, eng, actions
    torch.cuda.empty_cache()
    return rec, cfg


def _no_hook():
    pass


def run_tree_wave(torch, device, *, grid=200, roots=1024, sims=256, depth=5, root_steps=5, reps=4, wave=4):
    """BASELINE configs[4]: `roots` root states x `sims` simulations on a grid x grid map, every simulation descending
    `depth` levels with one covariance-only predict step per level (planning/mcts_zero/mcts.py:166-265), batched as
    one ipp_tree_step launch per level over (roots x sims-in-flight) items; node storage is recycled between waves.
    GRF ground truth (the reference's temperature dataset is not in the repo)."""
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.vec_env import cell_centre_actions

    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    # wave: simulations of one root in flight per launch (virtual-loss style batching): 4096 items per level
    n_items = roots * wave
    eng = IPPEngine(cfg, capacity=roots, state="factor", rank_cap=9 * (root_steps + depth + 1), window_rows=-1, fixed_prior=True,
                    node_capacity=n_items * depth, max_batch=n_items, device=device)
    white = torch.empty((roots, cfg.n_cells), dtype=torch.float32, device=device)
    eng.normal_rows(white, cfg.n_cells, 3, 1 << 40)
    eng.reset(white_noise=white)
    prev = torch.tensor([2.0, 2.0, 14.0], dtype=torch.float64, device=device).repeat(roots, 1)
    noise = torch.empty((root_steps, roots, eng.meas_cap), dtype=torch.float32, device=device)
    eng.normal_rows(noise, eng.meas_cap, 3, 2 << 40)
    for t in range(root_steps):  # the roots: a few executed steps each
        a = torch.as_tensor(cell_centre_actions(cfg, t, 0, roots, roots, ALTITUDES), device=device)
        eng.step(a, prev, meas_noise=noise[t])
        prev = a
    root_ids = torch.arange(roots, dtype=torch.int32, device=device).repeat_interleave(wave)
    prev_items = prev.repeat_interleave(wave, dim=0)
    # every simulation's actions stay near its root's last waypoint (a search explores the reachable neighbourhood)
    acts = [torch.as_tensor(cell_centre_actions(cfg, 100 + d, 0, n_items, n_items, ALTITUDES), device=device) for d in range(depth)]
    paths = torch.full((depth, n_items, 6), -1, dtype=torch.int32, device=device)
    new_ids = [(d * n_items + torch.arange(n_items, device=device)).to(torch.int32) for d in range(depth)]
    for d in range(1, depth):
        paths[d] = paths[d - 1]
        paths[d, :, d - 1] = new_ids[d - 1]
    reward = torch.empty(n_items, dtype=torch.float32, device=device)
    status = torch.empty(n_items, dtype=torch.int32, device=device)

    def one_wave():
        p = prev_items
        for d in range(depth):
            eng.tree_step(root_ids, paths[d], acts[d], p, new_ids=new_ids[d], reward_out=reward, status_out=status)
            p = acts[d]

    waves = sims // wave
    one_wave()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for _ in range(waves):
            one_wave()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = int(status.abs().sum().item()) == 0 and bool(torch.isfinite(reward).all().item())
    root_rank = float(eng.ranks().float().mean().item())
    n_steps = roots * sims * depth
    # roofline of the tree kernel: bytes counted on the device (stored rows of the chained state + the m new columns + the
    # node's diagonal per touched cell, SURVEY 8(d) cfg 5) / HIP-event duration of its launches, over 8 more waves
    eng.profile(True)
    eng.streamed_bytes(reset=True)
    for _ in range(8):
        one_wave()
    torch.cuda.synchronize()
    counted, _ = eng.streamed_bytes_detail(reset=True)
    k_ms, k_n = eng.profile_read(0)
    p_ms, p_n = eng.profile_read(2)  # launches >= 2048 items run as k_tree_prepare + k_tree_gain (else the fused k_tree_step)
    eng.profile(False)
    split = p_n > 0
    patch_tree = bool(int(eng.info.patch_layout))
    gbs = (counted / max(k_n, 1)) / ((k_ms + (p_ms if split else 0.0)) * 1e-3) / 1e9 if k_ms > 0 else 0.0
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return {"name": f"BASELINE configs[4]: {roots} roots x {sims} sims, {grid}x{grid} grid, depth {depth}, GRF ground truth "
                    f"(predict steps at tree nodes, ipp_tree_step; {wave} simulations per root per launch)",
            "value": n_steps / dt, "unit": "tree-steps/s", "ms_per_search": dt * 1e3, "root_rank": root_rank,
            "launch_items": n_items, "all_status_ok": ok,
            "kernel": "k_tree_patch" if patch_tree else ("k_tree_prepare + k_tree_gain" if split else "k_tree_step"),
            "kernel_ms_avg": k_ms + (p_ms if split else 0.0), "prepare_ms_avg": p_ms if split else None, "achieved_gbs": gbs,
            "frac": gbs / HBM_PEAK_GBS}


def run_mcts_driver(torch, device, *, grid=200, roots=1024, sims=256, in_flight=4, root_steps=3, driver="device"):
    """BASELINE configs[4] through the tree-search DRIVER (ipp_rl_amd/planning/mcts_zero/vector_mcts.py: PUCT selection,
    valid-action mask, forced playouts, Dirichlet noise, transposition-aware backup; reference
    planning/mcts_zero/mcts.py:83-296) with a stubbed network (uniform priors, constant value): 1024 roots x 256
    simulations on 200x200.  driver="device": selection, valid sets, expansion and backup in csrc/k_mcts.h (DeviceMCTS, one
    wavefront per root); driver="host": the same search with selection and bookkeeping in NumPy, vectorised over the roots
    (VectorMCTS; host-bound, kept for comparison).  Either way every covariance step runs on the device in ipp_tree_step
    launches shared by all roots (device driver: ONE launch per wave of simulations; host driver: one per tree level).
    seconds_per_search returns the policies in the reference's format (per-root dicts); seconds_per_search_device_policies leaves
    them on the device as [roots, kmax] arrays (get_policy(as_arrays=True))."""
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.planning.mcts_zero.device_mcts import DeviceMCTS
    from ipp_rl_amd.planning.mcts_zero.vector_mcts import VectorMCTS
    from ipp_rl_amd.vec_env import cell_centre_actions

    cfg = EngineConfig(x_dim=grid, y_dim=grid)
    horizon = 5
    eng = IPPEngine(cfg, capacity=roots, state="factor", rank_cap=9 * (root_steps + horizon + 2), window_rows=-1, fixed_prior=True,
                    node_capacity=roots * (sims + in_flight), max_batch=roots * in_flight, device=device)
    white = torch.empty((roots, cfg.n_cells), dtype=torch.float32, device=device)
    eng.normal_rows(white, cfg.n_cells, 9, 1 << 40)
    eng.reset(white_noise=white)
    prev = np.tile([2.0, 2.0, 14.0], (roots, 1))
    noise = torch.empty((root_steps, roots, eng.meas_cap), dtype=torch.float32, device=device)
    eng.normal_rows(noise, eng.meas_cap, 9, 2 << 40)
    for t in range(root_steps):
        a = cell_centre_actions(cfg, t, 0, roots, roots, [8.0, 14.0])
        eng.step(a, prev, meas_noise=noise[t])
        prev = a
    hyper = dict(gamma=1.0, puct_init=15.0, puct_base=10000.0, forced_playout_factor=2.0, max_valid_action_distance=11.5,
                 dirichlet_alpha=1.0, dirichlet_eps=0.25, num_mcts_simulations=sims)
    meta = {"budget": 100.0, "initial_budget": 100.0, "episode_horizon": horizon, "min_altitude": 8.0, "max_altitude": 14.0,
            "altitude_spacing": 6.0, "uav_specifications": {"max_v": 2.0, "max_a": 2.0},
            "scenario_info": {"value_threshold": 0.4, "interval_factor": 0}}
    if driver == "device":
        mcts = DeviceMCTS(eng, hyper, meta, None, sims_in_flight=in_flight, tie_break="random", leaf_value=0.3)
        mcts.get_policy(list(range(roots)), prev, [100.0] * roots)  # (first call allocates the search tables)
        mcts.stats.update(device_steps=0, launches=0, inferences=0)
    else:
        mcts = VectorMCTS(eng, hyper, meta, lambda reqs: [(None, 0.3)] * len(reqs), sims_in_flight=in_flight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = mcts.get_policy(list(range(roots)), prev, [100.0] * roots)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(o is not None and abs(sum(o[0].values() if isinstance(o[0], dict) else o[0]) - 1.0) < 1e-9 for o in out)
    st = dict(mcts.stats)
    dt_arrays = None
    if driver == "device":
        mcts.get_policy(list(range(roots)), prev, [100.0] * roots, as_arrays=True)  # (first call in this form: the result tensors are allocated)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        arr = mcts.get_policy(list(range(roots)), prev, [100.0] * roots, as_arrays=True)
        torch.cuda.synchronize()
        dt_arrays = time.perf_counter() - t0
        ok = ok and bool((arr["ok"] == 1).all().item()) and bool(((arr["policy"].sum(dim=1) - 1.0).abs() < 1e-9).all().item())
    eng.close()
    del eng, mcts
    torch.cuda.empty_cache()
    where = "PUCT / valid sets / expansion / backup on the device (k_mcts.h)" if driver == "device" else \
        "PUCT / backup on the host, vectorised over the roots (host-bound)"
    return {"name": f"BASELINE configs[4] through the tree-search driver: {roots} roots x {sims} simulations ({in_flight} in flight per "
                    f"root), {grid}x{grid}, horizon {horizon}, stubbed network; {where}",
            "value": roots * sims / dt, "unit": "simulations/s", "seconds_per_search": dt, "seconds_per_search_device_policies": dt_arrays,
            "device_tree_steps": st["device_steps"],
            "launches": st["launches"], "nodes": st["nodes"], "inferences": st["inferences"], "all_policies_valid": ok}


def schedule_text(rec):
    if rec["parts"] > 1:
        return (f"{rec['parts']} fixed groups of envs, one launch and one stream per group (VecIPPEnv.step_async -> ipp_step_parts): a group's "
                "step t + 1 is ordered behind its OWN step t only (envs are independent), so consecutive steps of different groups overlap "
                "on the device; every env still makes exactly `steps` steps inside the timed region")
    return "one launch per step on one stream"


def kernel_ms_text(rec):
    if rec["parts"] > 1:
        return ("per STEP: time during which at least one of the step's launches runs (union of the dispatches' HIP-event intervals over a "
                "leg of `steps` steps on the partitioned schedule, with an event pair around every dispatch -- which costs the overlap a few us per "
                "step, so this can exceed ms_per_step of the timed regions, which run without events; tools/trace_overlap.py gives the same union "
                "from the rocprofv3 kernel trace, profiles/r04_trace_overlap.txt).  The groups' launches overlap: their individual durations "
                "(part_launch_ms_avg, what rocprofv3 --stats averages together with the one-launch legs) add up to more than the wall time.  "
                "single_launch_ms_avg / single_launch_frac: the whole batch as ONE launch, alone on the device (the figure of rounds 1-3)")
    return "average HIP-event duration of the step kernel's dispatches over the roofline leg"


def sync_record(rec, total_envs, steps):
    if not rec.get("sync_elapsed_max_s"):
        return None
    return {"schedule": "one launch per step on one stream (rounds 1-3)", "value": aggregate_rate(total_envs, steps, rec["sync_elapsed_max_s"]),
            "ms_per_step": 1e3 * rec["sync_elapsed_max_s"] / steps,
            "region_ms_per_step": [1e3 * t / steps for t in rec["sync_region_elapsed_max_s"]]}


def extra_record(name, rec, total_envs):
    return {"name": name, "value": aggregate_rate(total_envs, rec["steps"], rec["elapsed_max_s"]), "unit": "env-steps/s",
            "ms_per_step": 1e3 * rec["elapsed_max_s"] / rec["steps"], "kernel": rec["kernel"], "kernel_ms_avg": rec["kernel_ms"],
            "achieved_gbs": rec["achieved_gbs"], "frac": rec["achieved_gbs"] / HBM_PEAK_GBS,
            "step_frac": rec["bytes_per_launch"] / (rec["elapsed_max_s"] / rec["steps"]) / 1e9 / HBM_PEAK_GBS,
            "frac_definition": "frac: algorithmic bytes / the dominant kernel's average duration; step_frac: the same bytes / the whole step period "
                               "(every launch of the step, resets and ground-truth generation included)",
            "necessary_bytes_per_launch": rec["necessary_bytes_per_launch"],
            "frac_necessary": (rec["necessary_bytes_per_launch"] / (rec["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if rec["necessary_bytes_per_launch"] and rec["kernel_ms"] else None,
            "window_rows": rec["window_rows"], "schedule_parts": rec["parts"], "single_launch_ms_avg": rec["single_launch_ms"],
            "sync_schedule": sync_record(rec, total_envs, rec["steps"]),
            "mean_rank_after_step": rec["mean_rank_after_step"], "arena_gb": rec["arena_gb"],
            "arena_kind": rec["arena_kind"], "arena_chunk_mib": rec["arena_chunk_mib"],
            "items_with_nonzero_status": rec["bad_status"], "non_finite_rewards": rec["bad_rewards"]}


def _config_block(args, rec, B, T, total_envs, pinning):
    """`config` of the line.  ORDER MATTERS: the driver's parser keeps the first ~20 keys, so the scalars that say whether a run is
    valid and where its streams / arena / ranks landed come first, prose and lists last."""
    q = rec["queues"] or {}
    issue = rec.get("issue_ms_per_rank") or []
    per_rank = rec["per_rank_s"]
    own = rec.get("own_ms_per_rank") or []
    slowest = max(range(len(own)), key=lambda r: own[r]) if own else 0  # (by the time until a rank's OWN device was done)
    regs = rec["region_elapsed_max_s"]
    head = {
        "workload": f"{baseline_config_name(args.grid, B, T, total_envs)}: {B} parallel envs per GPU, {args.grid}x{args.grid} grid, 10 altitude "
                    f"levels 5-14 m, example.yaml sensor/prior/UAV, GRF ground truth, adaptive reward with "
                    f"flight-time cost; {'predict-only (reward) calls' if args.predict_only else 'full fused env step (predict + observe + update)'}",
        "envs_per_gpu": B, "grid": f"{args.grid}x{args.grid}", "episode_steps": T, "window_rows": rec["window_rows"],
        "schedule_parts": rec["parts"],
        "arena_kind": rec["arena_kind"], "arena_chunk_mib": rec["arena_chunk_mib"],
        "queues_n_queues": q.get("n_queues"), "queues_parts_distinct": q.get("parts_distinct"),
        "queues_staging_shares_a_part_queue": q.get("staging_shares_a_part_queue"),
        "host_issue_ms_per_step": issue[0] if issue else None,
        "host_issue_ms_min_over_ranks": min(issue) if issue else None, "host_issue_ms_max_over_ranks": max(issue) if issue else None,
        "slowest_rank": slowest,
        "region_ms_min": 1e3 * min(regs) / args.steps, "region_ms_max": 1e3 * max(regs) / args.steps,
        "items_with_nonzero_status": rec["bad_status"], "non_finite_rewards": rec["bad_rewards"],
        "ranks_pinned": bool(pinning.get("pinned")),
    }
    rest = {
        "envs_total": total_envs, "state_repr": args.state, "mean_rank_after_step": rec["mean_rank_after_step"],
        "tile_threads": rec["workgroup_threads"],
        "prior": "shuffled per episode (window sized for 1.2 l)" if args.shuffle_prior else "fixed (example.yaml)",
        "rng": "device Philox4x32-10 keyed on the global env id", "episodes": "staggered (stationary rank mix)",
        "timed_regions": len(regs), "value_is": "median timed region",
        "region_ms_first": 1e3 * regs[0] / args.steps,
        "region_ms_per_step": [1e3 * t / args.steps for t in regs],
        "region_host_issue_ms_per_step": [1e3 * t / args.steps for t in rec["region_issue_s"]],
        "per_rank_ms_per_step": [1e3 * t / args.steps for t in per_rank],
        "per_rank_env_steps_per_s": [B * args.steps / t for t in per_rank],
        "host_issue_ms_per_rank": issue, "own_ms_per_step_per_rank": own,
        "rank_pinning": pinning,
        "split_min_items": rec["split_min_items"],
        **{"queues_" + k: v for k, v in q.items() if k not in ("n_queues", "parts_distinct", "staging_shares_a_part_queue")},
        "schedule": schedule_text(rec),
        "sync_schedule": sync_record(rec, total_envs, args.steps),
    }
    return {**head, **rest}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.print_args:
        print(json.dumps(workload_key(args)))
        return
    maybe_self_launch(args, argv)  # N > 1 without a launcher: become the launcher (before any GPU call)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC (RCCL)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pinning = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))  # before torch / HIP start threads
    import torch

    # Test hook for single-GPU boxes (tests/test_hip_bench_launcher.py): IPP_BENCH_SHARE_GPU=1 puts every rank on cuda:0
    # and uses gloo (RCCL refuses two ranks on one device), so that the launcher, the rank plumbing, the shard offsets and
    # the aggregation run end to end; the numbers of such a run mean nothing.
    share = os.environ.get("IPP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    ranks = Ranks("gloo" if share else "nccl", device=None if share else device)
    rank, world = ranks.rank, ranks.world
    lo, hi, total_envs, scaling = shard_plan(args, rank, world)
    B, T = hi - lo, args.episode_steps

    import contextlib

    prio_ctx = torch.cuda.stream(torch.cuda.Stream(device=device, priority=args.step_priority)) if args.step_priority else contextlib.nullcontext()
    with prio_ctx:
        rec, cfg = run_env_workload(torch, ranks, device, grid=args.grid, envs_local=B, env_lo=lo, total_envs=total_envs,
                                    episode_steps=T, state=args.state, window_rows=args.window_rows,
                                    shuffle_prior=args.shuffle_prior, tile_threads=args.tile_threads,
                                    predict_only=args.predict_only, fused_resets=args.fused_resets, steps=args.steps,
                                    warmup=args.warmup, regions=args.regions, parts=args.parts)
    if rank == 0:
        traffic, traffic_source = pmc_traffic(rec["kernel"], workload_key(args))
        out = {
            "metric": "env-steps/s (batched) on 50x50 grid" if args.grid == 50 else f"env-steps/s (batched) on {args.grid}x{args.grid} grid",
            "value": aggregate_rate(total_envs, args.steps, rec["elapsed_max_s"]),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * rec["elapsed_max_s"] / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": _config_block(args, rec, B, T, total_envs, pinning),
            "roofline": {
                "bound": "hbm", "achieved": rec["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rec["achieved_gbs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "traffic_definition": "HBM bytes per step (2 x FETCH_SIZE + WRITE_SIZE, KiB) of the whole batch run as one launch (--parts 1 under --pmc)",
                "traffic_over_algorithmic": (traffic / rec["bytes_per_launch"]) if traffic and rec["bytes_per_launch"] else None,
                # the same bytes over the STEP PERIOD of the timed regions (no events attached; every launch of the step, resets and
                # ground-truth generation inside): kernel <= step holds for this pair, the event leg's kernel_ms_avg can exceed ms_per_step
                "frac_step_clock": rec["bytes_per_launch"] / (rec["elapsed_max_s"] / args.steps) / 1e9 / HBM_PEAK_GBS,
                "step_frac": rec["bytes_per_launch"] / (rec["elapsed_max_s"] / args.steps) / 1e9 / HBM_PEAK_GBS,
                "kernel": rec["kernel"], "kernel_ms_avg": rec["kernel_ms"], "kernel_ms_avg_is": "event leg (a HIP-event pair on every dispatch, separate from the timed regions)",
                "launches": rec["launches"],
                "kernel_ms_definition": kernel_ms_text(rec), "launches_per_step": rec["parts"],
                "single_launch_ms_avg": rec["single_launch_ms"],
                "single_launch_frac": (rec["bytes_per_launch"] / (rec["single_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if rec["single_launch_ms"] else None,
                "part_launch_ms_avg": rec["part_launch_ms_avg"], "part_launches": rec["part_launches"],
                # one group's launch on its own: its share of the step's bytes / its own duration -- while the other group's launch runs
                # beside it (two of these overlap; the step-level figure above is bytes / the union of their intervals)
                "part_launch_frac_each": (rec["bytes_per_launch"] / rec["parts"] / (rec["part_launch_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                         if rec["part_launch_ms_avg"] else None,
                "algorithmic_bytes_per_launch": rec["bytes_per_launch"],
                "necessary_bytes_per_launch": rec["necessary_bytes_per_launch"],
                "frac_necessary": (rec["necessary_bytes_per_launch"] / (rec["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                  if rec["necessary_bytes_per_launch"] and rec["kernel_ms"] else None,
                "traffic_over_necessary": (traffic / rec["necessary_bytes_per_launch"]) if traffic and rec["necessary_bytes_per_launch"] else None,
                "necessary_bytes_definition": "the same count with every stored row charged only on the cells INSIDE that column's own rectangle "
                                              "(the lanes outside are masked requests that fetch nothing): sum over rows of popcount(lanes in the rectangle) x 8 "
                                              "+ (m + 4) x 4 bytes per cell of the new rectangle, counted on the device next to algorithmic_bytes",
                "algorithmic_bytes_definition": "SURVEY 8(d): 4 x (stored rows + m new rows + 4) bytes per touched cell, counted on the device: "
                                                "stored rows = the columns that contribute to this step (non-zero row of H U^T), touched cells = the "
                                                "cells of the rectangle within window_rows of the footprint in BOTH directions (a stored column IS that "
                                                "rectangle, as a compact patch; the padding columns a patch row may have are neither counted nor read)",
                "note": "columns of U are stored as compact patches of their rectangles (k_step_patch.h + k_patch_units.h): one 3-wave workgroup per item, "
                        "units of 128 valid cells, active columns as bit masks walked with scalar instructions.  What bounds a step at 4096 items "
                        "(profiles/r04_experiments.txt 12-13, per-item timeline of tools/timeline_parts.py): items hold a workgroup slot for 31.6 us on "
                        "average (prologue 13 us of dependent loads + fp64 header + m x m algebra, units 19 us), 2048 slots -> 63 us per step if no slot "
                        "ever idled; one launch per step idles them behind its stragglers (a launch lasts as long as its longest item, 65-75 us, and the "
                        "second round of items starts at 40-50 us: 89 us), two groups on two queues overlap one group's stragglers with the other's start "
                        "(73 us; each group's launch still lasts 67 us = its longest item)",
                "bytes_per_launch_incl_mask_reread": rec["bytes_per_launch"] + rec["mask_reread_bytes_per_launch"],
                "full_column_formula_bytes_per_launch": rec["formula_bytes_per_launch"],
                "other_kernels_ms_avg": rec["other_kernels_ms_avg"],
            },
        }
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only: at N > 1 the ranks are pinned to their share of the cores)
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, args)
            except Exception as exc:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
    # ---- the other configs that fit one GPU, same measurement, short runs (N = 1 default line only)
    default_line = (world == 1 and not args.no_extra and args.grid == 50 and B == 4096 and T == 40 and args.state == "factor"
                    and args.window_rows == -1 and not args.predict_only and not args.shuffle_prior and args.tile_threads == 0)
    if default_line:
        extra = []
        # (the 35-GB arena first: its 1-GiB physical chunks come out of device memory that nothing has carved up yet -- behind configs[2],
        # whose gigabytes of staging tensors go back to the driver in between, the same workload ran at 0.55-0.57 instead of 0.50 ms per step,
        # profiles/r06_experiments.txt 16)
        todo = [
            ("BASELINE configs[3] per-GPU share: 32768 envs, 50x50 grid, 40-step episodes",
             dict(grid=50, envs_local=32768, episode_steps=40)),
            ("BASELINE configs[2]: 32768 envs, 100x100 grid, 16-step episodes",
             dict(grid=100, envs_local=32768, episode_steps=16)),
            ("BASELINE configs[1], window 12 rows (valid for shuffle_prior_cov, the reference's self-play prior)",
             dict(grid=50, envs_local=4096, episode_steps=40, shuffle_prior=True)),
            ("BASELINE configs[1], predict-only calls (simulate_prediction_step, no state write)",
             dict(grid=50, envs_local=4096, episode_steps=40, predict_only=True)),
            ("BASELINE configs[1] with one launch per step on one stream (the schedule of rounds 1-3; frac: bytes / that launch's duration)",
             dict(grid=50, envs_local=4096, episode_steps=40, parts=1, steps=80, warmup=8)),
        ]
        for name, kw in todo:
            try:
                # (three regions, the median one reported: a single 20-step region now and then catches a 40-ms stall right behind the
                # release of the previous workload's arena -- the driver clears freed device memory in the background)
                kw = dict(dict(steps=20, warmup=4, parts=args.parts, regions=3), **kw)
                r, _ = run_env_workload(torch, ranks, device, env_lo=0, total_envs=kw["envs_local"], **kw)
                extra.append(extra_record(name, r, kw["envs_local"]))
            except Exception as exc:
                extra.append({"name": name, "error": repr(exc)})
        for w in (4, 8):  # (simulations of a root per launch: 4096 / 8192 items, as the device search has them in flight)
            try:
                extra.append(run_tree_wave(torch, device, wave=w))
            except Exception as exc:
                extra.append({"name": "BASELINE configs[4] tree wave", "error": repr(exc)})
        # (4 simulations in flight per root is what the parity tests pin; 8 halves the launches: the virtual visits keep the
        # descents apart either way, and 1 reproduces the reference's sequential search)
        for drv, w in (("device", 4), ("device", 8), ("host", 4)):
            try:
                extra.append(run_mcts_driver(torch, device, driver=drv, in_flight=w))
            except Exception as exc:
                extra.append({"name": f"tree-search driver ({drv})", "error": repr(exc)})
        out["extra"] = extra
    if rank == 0:
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
