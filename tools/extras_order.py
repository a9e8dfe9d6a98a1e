"""Does a workload's step time depend on what ran before it in the process (physical memory handed back by earlier arenas)?
configs[2] / configs[3] share first in a fresh process, then behind the headline workloads as bench.py's extras run them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ranks = bench.Ranks("nccl", device="cuda:0")
dev = "cuda:0"
def run(name, **kw):
    kw = dict(dict(steps=20, warmup=4, parts=2), **kw)
    r, _ = bench.run_env_workload(torch, ranks, dev, env_lo=0, total_envs=kw["envs_local"], **kw)
    free, total = torch.cuda.mem_get_info()
    print(f"{name:34s} {1e3 * r['elapsed_max_s'] / r['steps']:.4f} ms per step   kernel {r['kernel_ms']:.4f}   arena {r['arena_kind']} {r['arena_gb']:.1f} GB   free {free / 2**30:.0f} GiB", flush=True)
order = sys.argv[1] if len(sys.argv) > 1 else "big-first"
big = [("configs[2]", dict(grid=100, envs_local=32768, episode_steps=16)), ("configs[3] share", dict(grid=50, envs_local=32768, episode_steps=40))]
small = [("headline", dict(grid=50, envs_local=4096, episode_steps=40)), ("window 12", dict(grid=50, envs_local=4096, episode_steps=40, shuffle_prior=True)),
         ("predict-only", dict(grid=50, envs_local=4096, episode_steps=40, predict_only=True)), ("one launch", dict(grid=50, envs_local=4096, episode_steps=40, parts=1))]
seq = (big + small + big) if order == "big-first" else (small + big + big)
for name, kw in seq:
    run(name, **kw)
