"""Arena placement (include/ipp_engine.h "Arena placement", ipp-rl_amd/csrc/ipp_arena.hip): arenas straight from the driver give
the same results as a torch tensor, are owned / not owned as documented, and the probes run on them."""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu


def _episode(arena, dim=40, B=64, steps=6):
    import torch
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

    cfg = EngineConfig(x_dim=dim, y_dim=dim)
    env = VecIPPEnv(cfg, B, episode_steps=4, stagger=True, window_rows=-1, seed=5, parts=1, arena=arena)
    env.reset()
    rewards = []
    for t in range(steps):
        a = torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, [float(x) for x in range(5, 15)]), device="cuda")
        r, s = env.step(a)
        assert int(s.abs().sum()) == 0
        rewards.append(r.clone())
    torch.cuda.synchronize()
    out = (torch.stack(rewards).cpu().numpy(), env.mean(3).cpu().numpy(), env.diag(B - 1).cpu().numpy(), env.engine.arena_kind)
    env.close()
    return out


def test_driver_arenas_give_the_torch_arena_s_results_bit_for_bit():
    ref = _episode("torch")
    assert ref[3] == "torch"
    for kind in ("hip", "vmm"):
        got = _episode(kind)
        assert got[3] == kind
        for a, b in zip(ref[:3], got[:3]):
            assert np.array_equal(a, b)


def test_vmm_arena_chunks_tail_alignment_and_reuse():
    import torch
    from ipp_rl_amd.engine import DeviceArena

    DeviceArena.trim()  # (chunks pooled by earlier tests)
    free0 = torch.cuda.mem_get_info()[0]
    # 2 whole chunks of 64 MiB + a tail of 5 MiB + 1 byte -> 3 x the 2-MiB granularity (or whatever the device recommends)
    n = 2 * (64 << 20) + (5 << 20) + 1
    a = DeviceArena(n, 0, kind="vmm", chunk_bytes=64 << 20, align_bytes=64 << 20)
    assert a.data_ptr() % (64 << 20) == 0
    a0 = a.data_ptr()
    t = a.as_tensor("cuda:0")
    assert t.numel() == n
    t.fill_(7)
    assert int(t[0]) == 7 and int(t[-1]) == 7 and int(t[64 << 20]) == 7 and int(t[2 * (64 << 20) + 17]) == 7
    used = free0 - torch.cuda.mem_get_info()[0]
    assert n <= used <= n + (64 << 20)  # the tail is not rounded up to a whole chunk
    ms = a.probe(items=64, rows=16, launches=2)
    ns = a.latency(waves=64, hops=200)
    assert 0.0 < ms < 50.0 and 100.0 < ns < 1e5
    del t
    a.free()
    a.free()  # idempotent
    # the two whole chunks stay in the library's pool for the next arena of this chunk size (the tail chunk went back) ...
    held = free0 - torch.cuda.mem_get_info()[0]
    assert (128 << 20) <= held < (128 << 20) + (8 << 20)
    b = DeviceArena(3 * (64 << 20), 0, kind="vmm", chunk_bytes=64 << 20, align_bytes=64 << 20)  # two pooled chunks + one new
    assert b.data_ptr() != a0 and b.data_ptr() % (64 << 20) == 0  # (never the address range of a freed arena)
    assert free0 - torch.cuda.mem_get_info()[0] < 3 * (64 << 20) + (8 << 20)
    tb = b.as_tensor("cuda:0"); tb.fill_(3); assert int(tb[-1]) == 3; del tb
    b.free()
    # ... until it is trimmed
    assert DeviceArena.trim(0) == 3 * (64 << 20) and DeviceArena.trim() == 0
    assert free0 - torch.cuda.mem_get_info()[0] < (8 << 20)
    with pytest.raises(ValueError):
        DeviceArena(1 << 20, 0, kind="nope")


def test_caller_owned_arena_survives_the_engine_and_a_short_one_is_refused():
    import torch
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd.engine import DeviceArena

    cfg = EngineConfig(x_dim=20, y_dim=20)
    probe = IPPEngine(cfg, capacity=4, state="factor", rank_cap=36, arena="torch")
    need = probe.arena_bytes
    probe.close()
    a = DeviceArena(need, 0, kind="hip")
    eng = IPPEngine(cfg, capacity=4, state="factor", rank_cap=36, arena=a)
    assert eng.arena_kind == "caller"
    eng.reset(white_noise=np.random.RandomState(0).normal(size=(4, 20, 20)))
    eng.close()
    assert a.data_ptr()  # still mapped: a second engine takes the same memory
    eng2 = IPPEngine(cfg, capacity=4, state="factor", rank_cap=36, arena=a)
    eng2.close()
    a.free()
    small = DeviceArena(4096, 0, kind="hip")
    with pytest.raises(ValueError):
        IPPEngine(cfg, capacity=4, state="factor", rank_cap=36, arena=small)
    small.free()


def test_auto_takes_the_virtual_memory_api_from_the_threshold(monkeypatch):
    from ipp_rl_amd import EngineConfig, IPPEngine
    from ipp_rl_amd import engine as engine_mod

    cfg = EngineConfig(x_dim=20, y_dim=20)
    e = IPPEngine(cfg, capacity=4, state="factor", rank_cap=36)
    assert e.arena_kind == "torch"  # small arenas stay with the allocator
    e.close()
    monkeypatch.setattr(engine_mod, "ARENA_VMM_MIN_BYTES", 1 << 16)
    e = IPPEngine(cfg, capacity=4, state="factor", rank_cap=36)
    assert e.arena_kind == "vmm" and e.arena.data_ptr() % (64 << 20) == 0
    e.reset(white_noise=np.random.RandomState(0).normal(size=(4, 20, 20)))
    e.close()
