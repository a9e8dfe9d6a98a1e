"""
`python bench.py --gpus 2` end to end on the one GPU of the test box: the parent process (which never touches the GPU)
starts two ranks through torch.distributed.run, each builds its shard's batched env at its global env-id offset, the
timed region is bracketed by barriers, rank 0 prints ONE JSON line with n_gpus = 2 and per-rank rates.  Both ranks share
cuda:0 and talk over gloo (IPP_BENCH_SHARE_GPU: RCCL refuses two ranks on one device), so only the plumbing is under
test here -- the RCCL path itself differs by the backend string.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks_and_reports_the_whole_job():
    env = dict(os.environ, IPP_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 2
    cfg = d["config"]
    assert cfg["envs_per_gpu"] == 256 and cfg["envs_total"] == 512
    assert len(cfg["per_rank_ms_per_step"]) == 2 and len(cfg["per_rank_env_steps_per_s"]) == 2
    assert cfg["items_with_nonzero_status"] == 0 and cfg["non_finite_rewards"] == 0
    # what a parser that keeps the first 20 keys of `config` still sees: validity, streams, arena, the ranks' host loops
    head = list(cfg)[:20]
    for k in ("workload", "envs_per_gpu", "arena_kind", "queues_n_queues", "queues_parts_distinct", "queues_staging_shares_a_part_queue",
              "host_issue_ms_per_step", "host_issue_ms_min_over_ranks", "host_issue_ms_max_over_ranks", "slowest_rank", "region_ms_min",
              "region_ms_max", "items_with_nonzero_status", "non_finite_rewards", "ranks_pinned"):
        assert k in head, (k, head)
    assert len(cfg["host_issue_ms_per_rank"]) == 2 and cfg["slowest_rank"] in (0, 1)
    assert cfg["host_issue_ms_min_over_ranks"] <= cfg["host_issue_ms_max_over_ranks"]
    assert cfg["slowest_rank"] == max(range(2), key=lambda r: cfg["own_ms_per_step_per_rank"][r])
    pin = cfg["rank_pinning"]
    assert pin["pinned"] == ((os.cpu_count() or 1) >= 2) and (not pin["pinned"] or pin["cores"] >= 1)
    assert d["roofline"]["frac_step_clock"] == d["roofline"]["step_frac"] and "event leg" in d["roofline"]["kernel_ms_avg_is"]
    # value = envs of ALL ranks x steps / the slowest rank's time
    assert abs(d["value"] - 512 * 6 / (max(cfg["per_rank_ms_per_step"]) * 6e-3)) / d["value"] < 1e-6
    assert d["roofline"]["kernel"] in ("k_step_patch", "k_step_factor") and d["roofline"]["achieved"] > 0
    # strong scaling: the total split into contiguous ranges
    cmd = cmd[:2] + ["--gpus", "2", "--envs-total", "384", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["envs_total"] == 384 and d["config"]["envs_per_gpu"] == 192
