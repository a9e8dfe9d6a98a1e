"""
World-size-2 gloo test of bench.py's multi-process logic on CPU: contiguous env-id shards, inputs derived from the
global env id (so the union over ranks equals the single-process workload), barrier + MAX-over-ranks timing,
aggregate = sum of per-rank units / max time.  No data-path collective exists (envs are independent).
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions

    cfg = EngineConfig(x_dim=50, y_dim=50)
    per_gpu, total = 16, 16 * world
    lo, hi = rank * per_gpu, (rank + 1) * per_gpu
    acts = torch.as_tensor(cell_centre_actions(cfg, 7, lo, hi, total, list(range(5, 15))))
    gathered = [torch.empty_like(acts) for _ in range(world)]
    dist.all_gather(gathered, acts)
    dist.barrier()
    elapsed = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)  # rank 1 is the slow one
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), torch.cat(gathered).numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), elapsed.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing(tmp_path):
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = cell_centre_actions(EngineConfig(x_dim=50, y_dim=50), 7, 0, 32, 32, list(range(5, 15)))
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), full)
    tmax = float(np.load(tmp_path / "tmax.npy")[0])
    assert abs(tmax - 0.020) < 1e-12
    value = 32 * 5 / tmax  # whole-job units / max-over-ranks time
    assert abs(value - 8000.0) < 1e-6
