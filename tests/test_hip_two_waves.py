"""The launch-size rule of the patch step kernel (csrc/ipp_engine.hip launch_chunk, ipp_info.patch_two_wave_min_items): large launches
run two waves per item (k_step_patch<2>, from 16384 items k_step_patch<2, 4, 6>), small ones three (k_step_patch<3>).  Same arithmetic
per cell in the same order (mapping/mappings.py:178-197; planning/common/rewards.py:8-31), so which instantiation a launch took must
not show in a single bit -- rewards, planes, ranks, ground truths -- through scheduled resets, on one launch per step and on two groups."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALTS = [float(a) for a in range(5, 15)]


def _run(monkeypatch, env_vars, dim, B, T, parts, steps):
    import torch
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)
    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=3, parts=parts)
    info = (int(env.engine.info.patch_waves), int(env.engine.info.patch_two_wave_min_items), int(env.engine.info.patch_big_min_items))
    for k in env_vars:
        monkeypatch.delenv(k)
    env.reset()
    rewards = []
    for t in range(steps):
        a = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda")
        r, s = env.step(a)
        assert int(s.abs().sum()) == 0
        rewards.append(r.clone())
    # ... and a predict-only call on the final states (Mapping.update_grid_map(predict_only=True), mapping/mappings.py:114: nothing is written)
    a = torch.as_tensor(cell_centre_actions(cfg, steps, 0, B, B, ALTS), device="cuda")
    rp, sp = env.engine.step(a, env.prev, predict_only=True, cov_only=True)
    assert int(sp.abs().sum()) == 0
    rewards.append(rp.clone())
    torch.cuda.synchronize()
    out = dict(rewards=torch.stack(rewards), ranks=env.engine.ranks().clone(), prev=env.prev.clone(),
               planes=[(env.mean(e).clone(), env.diag(e).clone(), env.ground_truth(e).clone()) for e in (0, 1, B // 2, B - 1)],
               cov=env.engine.read_cov(B // 3).clone() if dim <= 50 else None)
    env.close()
    return info, out


@pytest.mark.parametrize("dim,B,T,parts", [(50, 512, 12, 1), (50, 768, 12, 2), (100, 256, 6, 1)])
def test_two_wave_and_three_wave_launches_are_bit_identical(monkeypatch, dim, B, T, parts):
    import torch

    steps = 2 * T + 3
    i3, three = _run(monkeypatch, {"IPP_PATCH_TWO_MIN": "0"}, dim, B, T, parts, steps)                           # k_step_patch<3> always
    i2, two = _run(monkeypatch, {"IPP_PATCH_TWO_MIN": "1", "IPP_PATCH_BIG": "0"}, dim, B, T, parts, steps)       # k_step_patch<2> always
    ib, big = _run(monkeypatch, {"IPP_PATCH_TWO_MIN": "1", "IPP_PATCH_BIG": "1"}, dim, B, T, parts, steps)       # k_step_patch<2, 4, 6> always
    assert i3 == (3, 0, 0) and i2 == (3, 1, 0) and ib == (3, 1, 1)
    for other in (two, big):
        assert torch.equal(three["rewards"], other["rewards"])
        assert torch.equal(three["ranks"], other["ranks"]) and torch.equal(three["prev"], other["prev"])
        for a, b in zip(three["planes"], other["planes"]):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
        if three["cov"] is not None:
            assert torch.equal(three["cov"], other["cov"])


def test_default_rule_and_its_switch(monkeypatch):
    from ipp_rl_amd import EngineConfig, IPPEngine

    cfg = EngineConfig(x_dim=50, y_dim=50)
    eng = IPPEngine(cfg, capacity=16, state="factor", rank_cap=90, window_rows=-1, fixed_prior=True)
    assert (int(eng.info.patch_waves), int(eng.info.patch_two_wave_min_items), int(eng.info.patch_big_min_items)) == (3, 6144, 16384)
    eng.close()
    monkeypatch.setenv("IPP_PATCH_TWO_MIN", "0")
    eng = IPPEngine(cfg, capacity=16, state="factor", rank_cap=90, window_rows=-1, fixed_prior=True)
    assert (int(eng.info.patch_two_wave_min_items), int(eng.info.patch_big_min_items)) == (0, 0)
    eng.close()
