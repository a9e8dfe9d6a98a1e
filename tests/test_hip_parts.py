"""
The partitioned schedule of a batched step (VecIPPEnv(parts > 1): one launch and one stream per fixed group of envs,
ipp_step_parts) against the single launch: envs are independent (an episode is a chain of
Mapping.update_grid_map calls on ONE map, mapping/mappings.py:114-153), so rewards, planes, ranks, previous waypoints
and episode counters have to agree BIT FOR BIT however the launches overlap -- through scheduled resets, the two
noise rings and the staged ground-truth blocks, on the patch kernel and on the band-tile fused kernel.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALTS = [float(a) for a in range(5, 15)]


def _pair(parts, B=4096, T=8, dim=50, window_rows=-1, **kw):
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv

    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    envs = [VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=window_rows, seed=11, parts=p, **kw) for p in (1, parts)]
    assert envs[0].parts == 1 and envs[1].parts == parts
    for env in envs:
        env.reset()
    return cfg, envs


def _same_state(a, b, envs_to_check):
    import torch

    assert torch.equal(a.engine.ranks(), b.engine.ranks())
    assert torch.equal(a.prev, b.prev)
    assert np.array_equal(a.episode, b.episode)
    for e in envs_to_check:
        assert torch.equal(a.mean(e), b.mean(e)), e
        assert torch.equal(a.diag(e), b.diag(e)), e
        assert torch.equal(a.ground_truth(e), b.ground_truth(e)), e


@pytest.mark.parametrize("parts,window_rows,tile_threads,shuffle", [(2, -1, 0, False), (3, -1, 0, False), (2, 12, 256, False), (2, 12, 0, True)])
def test_async_parts_equal_single_launch(parts, window_rows, tile_threads, shuffle):
    """Full-size batch (races and stale reads show at 4096 envs, not at 64), 5 episodes of 8 steps: > 2 noise rings, > 4 staged
    ground-truth blocks; the async steps are never joined inside the loop except every 7th step."""
    import torch
    from ipp_rl_amd.vec_env import cell_centre_actions

    B, T = 4096, 8
    # shuffle: per-episode priors (shuffle_prior_cov, window 12): the partitioned env installs them through the resets folded into
    # its step launches (ipp_set_reset_prior), the single-launch env through its separate reset launch
    cfg, (one, many) = _pair(parts, B=B, T=T, window_rows=window_rows, tile_threads=tile_threads, shuffle_prior_cov=shuffle)
    assert bool(many._fused_reset) and bool(one._fused_reset) == (not shuffle)
    assert int(many.engine.info.fused_step) == 1
    assert int(many.engine.info.patch_layout) == (1 if tile_threads == 0 else 0)  # (explicit tile_threads: the band-tile fused kernel)
    acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(5 * T)]
    torch.cuda.synchronize()
    rewards = []
    for t, a in enumerate(acts):
        r1, s1 = one.step(a)
        rewards.append((r1.clone(), s1.clone()))
        many.step_async(a, inputs_ready=True)
        if t % 7 == 6:
            many.wait()
            assert torch.equal(torch.nan_to_num(many.reward, nan=-7.0), torch.nan_to_num(rewards[-1][0], nan=-7.0)), t
            assert torch.equal(many.status, rewards[-1][1]), t
    many.wait()
    torch.cuda.synchronize()
    assert torch.equal(many.reward, rewards[-1][0])
    assert int((many.status != 0).sum()) == 0
    _same_state(one, many, (0, 1, 39, 40, 41, 79, 80, 2047, 4095))
    # every group holds every episode phase, the groups partition the batch
    all_envs = torch.cat([many.part_envs(p) for p in range(parts)]).sort().values
    assert torch.equal(all_envs, torch.arange(B, device="cuda"))
    for p in range(parts):
        ph = (many.part_envs(p) % T).unique()
        assert ph.numel() == T


def test_sync_step_on_a_partitioned_env_and_mixed_call_forms():
    """step() on a partitioned env joins the part streams every step; calls the async path cannot serve (env subsets,
    given noise, hooks) run on the caller's stream behind the part streams -- same results as the plain env throughout."""
    import torch
    from ipp_rl_amd.vec_env import cell_centre_actions

    B, T = 512, 6
    cfg, (one, many) = _pair(2, B=B, T=T)
    rs = np.random.RandomState(3)
    for t in range(4 * T):
        a = cell_centre_actions(cfg, t, 0, B, B, ALTS)
        if t % 5 == 3:  # a hook forces the single-launch form on both
            hits = []
            r1, _ = one.step(a, after_step_hook=lambda: hits.append(1))
            r2, _ = many.step(a, after_step_hook=lambda: hits.append(2))
            assert hits == [1, 2]
        elif t % 5 == 4:  # async issue, then a synchronous read
            r1, _ = one.step(a)
            many.step_async(torch.as_tensor(a, device="cuda"))
            many.wait()
            r2 = many.reward
        else:
            r1, _ = one.step(a)
            r2, _ = many.step(a)
        assert torch.equal(r1, r2), t
    _same_state(one, many, (0, 5, 6, 7, 255, 511))
    # a hand-made reset of a subset, then on
    ids = np.sort(rs.choice(B, size=17, replace=False)).astype(np.int32)
    for env in (one, many):
        env.reset(env_ids=ids)
    for t in range(4 * T, 6 * T):
        a = cell_centre_actions(cfg, t, 0, B, B, ALTS)
        r1, _ = one.step(a)
        r2, _ = many.step(a)
        assert torch.equal(r1, r2), t
    _same_state(one, many, [int(i) for i in ids[:4]] + [0, 511])


def test_step_parts_argument_checks():
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd._ffi import IppError

    cfg = EngineConfig(x_dim=50, y_dim=50)
    eng = IPPEngine(cfg, capacity=64, state="factor", rank_cap=90, window_rows=-1, fixed_prior=True)
    gts = torch.rand((64, cfg.n_cells), device="cuda")
    eng.reset(gt=gts)
    torch.cuda.synchronize()  # (step_parts runs on other streams and joins nothing)
    a = torch.tensor([[102.0, 98.0, 9.0]], dtype=torch.float64, device="cuda").repeat(64, 1).contiguous()
    prev = torch.tensor([[2.0, 2.0, 14.0]], dtype=torch.float64, device="cuda").repeat(64, 1).contiguous()
    nz = torch.zeros((64, 9), device="cuda")
    reward = torch.empty(64, device="cuda")
    status = torch.empty(64, dtype=torch.int32, device="cuda")
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    with pytest.raises(IppError):  # no dispatch order installed
        eng.step_parts(a, prev, nz, 4 | 8, reward, status, [0, 32, 64], sts)
    eng.set_item_order(torch.arange(64, dtype=torch.int32, device="cuda"))
    for bad in ([0, 64, 64], [0, 32, 63], [1, 32, 64]):
        with pytest.raises(IppError):
            eng.step_parts(a, prev, nz, 4 | 8, reward, status, bad, sts)
    with pytest.raises(IppError):  # predict-only parts do not move the UAVs (IPP_PREDICT_ONLY | IPP_UPDATE_PREV)
        eng.step_parts(a, prev, nz, 2 | 32, reward, status, [0, 32, 64], sts)
    # predict-only parts = the predict-only launch: rewards of the candidate actions, state untouched
    r0, s0 = eng.step(a, prev, meas_noise=nz, predict_only=True, cov_only=True)
    torch.cuda.synchronize()
    ranks0 = eng.ranks().clone()
    eng.step_parts(a, prev, None, 1 | 2 | 4 | 8, reward, status, [0, 32, 64], sts)  # (the flags of the call above)
    torch.cuda.synchronize()
    assert torch.equal(reward, r0) and torch.equal(status, s0) and torch.equal(eng.ranks(), ranks0)
    eng.step_parts(a, prev, nz, 4 | 8, reward, status, [0, 32, 64], sts)
    torch.cuda.synchronize()
    ref = IPPEngine(cfg, capacity=64, state="factor", rank_cap=90, window_rows=-1, fixed_prior=True)
    ref.reset(gt=gts)
    r2, s2 = ref.step(a, prev, meas_noise=nz)
    assert torch.equal(reward, r2) and torch.equal(status, s2)
    dense = IPPEngine(cfg, capacity=4, state="dense")
    with pytest.raises(IppError):
        dense.set_item_order(torch.arange(4, dtype=torch.int32, device="cuda"))
        dense.step_parts(a[:4], prev[:4], nz[:4], 4 | 8, reward[:4], status[:4], [0, 2, 4], sts)


def test_part_and_staging_streams_sit_on_different_hardware_queues():
    """ipp_probe_stream_pair: two chains of dependent 30-us launches take twice as long on ONE hardware queue as on two.
    VecIPPEnv(parts=2) classes new streams by queue with it and must end up with its two part streams and the
    ground-truth staging stream on pairwise different queues (on one queue a staging launch holds back a group's next
    step: 43-47 M env-steps/s instead of 55-56 M at configs[1], profiles/r04_experiments.txt item 12)."""
    import torch
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv

    cfg = EngineConfig(x_dim=50, y_dim=50)
    env = VecIPPEnv(cfg, 256, episode_steps=8, stagger=True, window_rows=-1, seed=5, parts=2)
    eng = env.engine
    launches = 12
    serial = launches * 0.030 * 2
    st = torch.cuda.Stream()
    t_same = eng.probe_stream_pair(st, st, launches)
    assert t_same > 0.9 * serial, t_same  # one stream: the 2 x 12 launches run one after the other
    assert env._queues is not None and env._queues["n_queues"] >= 2
    assert env.part_queues_distinct == (env._queues["n_queues"] >= 3)
    if env._queues["n_queues"] >= 4:  # the runtime's default: enough queues for both groups, the staging and the caller
        shared = env._queues["probes_ms_shared"]
        a, b, s = env.part_stream(0), env.part_stream(1), env._side
        for x, y in ((a, b), (a, s), (b, s), (a, torch.cuda.current_stream()), (b, torch.cuda.current_stream())):
            assert min(eng.probe_stream_pair(x, y, launches) for _ in range(2)) < shared, "two of the env's streams share a queue"
        # the placement as plain numbers (bench.py puts them on its line: config.queues_*)
        rep = env.queue_report
        assert rep["n_queues"] >= 4 and rep["parts_distinct"] is True and rep["staging_shares_a_part_queue"] is False
    # one launch per step: the staging stream is only kept off the caller's queue (one probe per candidate, no classification)
    single = VecIPPEnv(cfg, 64, episode_steps=8, stagger=True, window_rows=-1, seed=5, parts=1)
    assert single.queue_report is None and single._side is not None
    assert eng.probe_stream_pair(single._side, torch.cuda.current_stream(), launches) < 0.9 * serial
    with pytest.raises(Exception):
        eng.probe_stream_pair(st, st, 0)
