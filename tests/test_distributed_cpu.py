"""
World-size-2 gloo test of bench.py's OWN multi-process code on CPU (no GPU call is reached): `Ranks` (process group,
barrier, per-rank times, MAX over ranks), `shard_plan` (contiguous env-id ranges, weak and strong scaling),
`timed_region` and `aggregate_rate`, plus the `--gpus N` self-launch decision.  No data-path collective exists
(envs are independent); the workload inputs derive from the global env id, so the union over ranks equals the
single-process workload.
"""
import json
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions

    ranks = bench.Ranks("gloo")  # the class bench.main() builds with "nccl"
    assert (ranks.rank, ranks.world) == (rank, world)
    cfg = EngineConfig(x_dim=50, y_dim=50)
    weak = bench.shard_plan(bench.parse(["--envs", "16"]), rank, world)
    strong = bench.shard_plan(bench.parse(["--envs-total", "33"]), rank, world)
    lo, hi, total, _ = weak
    acts = cell_centre_actions(cfg, 7, lo, hi, total, bench.ALTITUDES)
    calls = []

    def run_steps(k):  # rank 1 is the slow one
        for _ in range(k):
            time.sleep(0.004 * (rank + 1))
            calls.append(1)

    # a choice that changes the number of collectives a rank enters (the schedule of the step) holds only if every rank makes it
    agree = [ranks.all_agree(True), ranks.all_agree(rank == 0), ranks.all_agree(False), ranks.all_agree(rank != world - 1)]
    gathered = ranks.gather(10.0 * rank + 1.0)
    per_rank, tmax = bench.timed_region(run_steps, 5, lambda: None, ranks)
    value = bench.aggregate_rate(total, 5, tmax)
    np.save(os.path.join(out_dir, f"acts{rank}.npy"), acts)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
        json.dump({"weak": weak, "strong": strong, "per_rank": per_rank, "tmax": tmax, "value": value, "calls": len(calls), "agree": agree, "gathered": gathered}, fh)
    ranks.close()


def test_two_rank_sharding_and_timing_through_bench_functions(tmp_path):
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions
    import bench

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    # shards: contiguous, disjoint, covering; weak = 16 per rank, strong = 33 split 16 / 17
    assert [o["weak"] for o in out] == [[0, 16, 32, "weak"], [16, 32, 32, "weak"]]
    assert [o["strong"] for o in out] == [[0, 16, 33, "strong"], [16, 33, 33, "strong"]]
    full = cell_centre_actions(EngineConfig(x_dim=50, y_dim=50), 7, 0, 32, 32, bench.ALTITUDES)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"acts{r}.npy") for r in range(world)]), full)
    # timing: exactly 5 steps each; every rank sees the same per-rank list; the max is rank 1's time (>= 5 x 8 ms);
    # rank 0's own time includes waiting at the closing barrier, so it is >= its own 20 ms of work
    for o in out:
        assert o["agree"] == [True, False, False, False] and o["gathered"] == [1.0, 11.0]
        assert o["calls"] == 5 and len(o["per_rank"]) == 2
        assert o["per_rank"] == out[0]["per_rank"] and o["tmax"] == out[0]["tmax"]
        assert o["tmax"] == max(o["per_rank"]) and o["tmax"] >= 0.040 and o["per_rank"][0] >= 0.020
        assert abs(o["value"] - 32 * 5 / o["tmax"]) < 1e-9


def test_self_launch_decision_and_command(monkeypatch):
    import bench

    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    args = bench.parse(argv)
    cmd = bench.launch_command(args, argv, 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-len(argv) - 1].endswith("bench.py") and cmd[-len(argv):] == argv
    seen = []

    class Done:
        returncode = 7

    def fake_run(c, env=None):
        seen.append((c, env))
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    # ranks started by a launcher (WORLD_SIZE set) and single-GPU runs never re-launch
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.maybe_self_launch(args, argv)
    monkeypatch.delenv("WORLD_SIZE")
    bench.maybe_self_launch(bench.parse([]), [])
    assert seen == []
    # N > 1 without a launcher: one child launcher, its exit code is ours; torch has not been imported by bench itself
    try:
        bench.maybe_self_launch(args, argv)
        raise AssertionError("expected SystemExit")
    except SystemExit as exc:
        assert exc.code == 7
    assert len(seen) == 1 and seen[0][0][:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert seen[0][1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_eight_ranks_one_disagreeing_rank_and_the_slowest_one(tmp_path):
    """World 8 (the node bench.py is written for): contiguous shards of configs[3]'s 262144 envs, a schedule choice on which ONE rank
    disagrees (its streams had to share a queue) takes every rank to the common fallback, every rank sees the same per-rank times and
    the job's time is the slowest rank's."""
    import bench

    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert [o["shard"][:2] for o in out] == [[32768 * r, 32768 * (r + 1)] for r in range(world)]
    assert all(o["shard"][2:] == [262144, "strong"] for o in out)
    for o in out:
        assert o["agree"] == [True, False, False]  # all / all but rank 5 / none
        assert o["per_rank"] == out[0]["per_rank"] and len(o["per_rank"]) == world
        assert o["tmax"] == max(o["per_rank"]) and o["tmax"] >= 0.036  # (every rank's time includes the wait for rank 3 at the barrier)
        for key in ("issue", "own"):  # ... which rank that was is in the times taken BEFORE the barrier: rank 3 sleeps longest
            assert o[key] == out[0][key] and len(o[key]) == world and max(range(world), key=lambda r: o[key][r]) == 3
            assert min(o[key]) < 0.5 * max(o[key])
        assert abs(o["value"] - 262144 * 3 / o["tmax"]) < 1e-6


def _worker8(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench

    ranks = bench.Ranks("gloo")
    shard = bench.shard_plan(bench.parse(["--envs-total", "262144", "--gpus", "8"]), rank, world)
    agree = [ranks.all_agree(True), ranks.all_agree(rank != 5), ranks.all_agree(False)]
    issue = []

    def run_steps(k):
        for _ in range(k):
            time.sleep(0.012 if rank == 3 else 0.002)

    per_rank, tmax = bench.timed_region(run_steps, 3, lambda: None, ranks, issue)
    gathered = ranks.gather(1e3 * issue[0][0] / 3)
    own = ranks.gather(1e3 * issue[0][1] / 3)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
        json.dump({"shard": shard, "agree": agree, "per_rank": per_rank, "tmax": tmax, "issue": gathered, "own": own,
                   "value": bench.aggregate_rate(shard[2], 3, tmax)}, fh)
    ranks.close()


def test_rank_core_plan_is_disjoint_and_follows_the_gpus_numa_nodes(tmp_path):
    import bench

    allowed = set(range(64))
    # no NUMA information: even contiguous split
    plans = [bench.plan_rank_cores(r, 8, allowed)[0] for r in range(8)]
    assert all(len(p) == 8 for p in plans) and sorted(c for p in plans for c in p) == list(range(64))
    # two sockets, hyper-thread siblings enumerated behind the physical cores; GPUs 0-3 on node 0, 4-7 on node 1
    node_cpus = {0: list(range(0, 16)) + list(range(32, 48)), 1: list(range(16, 32)) + list(range(48, 64))}
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [bench.plan_rank_cores(r, 8, allowed, gpu_nodes, node_cpus) for r in range(8)]
    assert sorted(c for p, _ in plans for c in p) == list(range(64))
    for r, (cores, how) in enumerate(plans):
        assert len(cores) == 8 and set(cores) <= set(node_cpus[gpu_nodes[r]]) and f"NUMA node {gpu_nodes[r]}" in how
    # a restricted affinity mask (a container with 8 CPUs), 8 ranks: one core each; fewer cores than ranks: no pinning plan
    assert [bench.plan_rank_cores(r, 8, set(range(8)))[0] for r in range(8)] == [[r] for r in range(8)]
    assert bench.plan_rank_cores(2, 8, {0, 1, 2})[0] == [0, 1, 2]
    assert bench.plan_rank_cores(0, 1, allowed)[0] == sorted(allowed)
    # unknown node of one GPU: fall back to the even split for everybody (disjoint either way)
    plans = [bench.plan_rank_cores(r, 8, allowed, [0, 0, 0, -1, 1, 1, 1, 1], node_cpus)[0] for r in range(8)]
    assert sorted(c for p in plans for c in p) == list(range(64))
    # sysfs reader on a fake tree: two AMD GPUs (class 0x0380 / 0x1200), one other vendor, one AMD non-GPU function
    for i, (vendor, cls, node) in enumerate([("0x1002", "0x038000", 0), ("0x10de", "0x030000", 1), ("0x1002", "0x120000", 1), ("0x1002", "0x060400", 0)]):
        d = tmp_path / "bus" / "pci" / "devices" / f"0000:0{i}:00.0"
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n"); (d / "class").write_text(cls + "\n"); (d / "numa_node").write_text(f"{node}\n")
    assert bench.gpu_numa_nodes(str(tmp_path)) == [0, 1]
    assert bench._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
