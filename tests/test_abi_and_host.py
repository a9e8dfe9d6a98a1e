"""
CPU-only checks: the C-ABI library loads and exports every symbol include/ipp_engine.h declares (no compute
without a GPU), the ctypes binding mirrors the header, the product path fails loudly without a GPU, and the
host-side mirrors of the reference's scalar helpers agree with golden vectors recorded from the reference.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.params import example_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "ipp_engine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ipp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from ipp_rl_amd import _ffi

    assert os.path.exists(_ffi.LIB_PATH), "build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ipp_engine.h but not exported"
    assert sorted(_ffi.PROTOTYPES) == names, "ctypes prototypes and header declarations differ"
    assert _ffi.load().ipp_abi_version() == _ffi.ABI_VERSION
    txt = open(os.path.join(ROOT, "include", "ipp_engine.h")).read()
    for macro, val in (("IPP_COV_ONLY", _ffi.IPP_COV_ONLY), ("IPP_PREDICT_ONLY", _ffi.IPP_PREDICT_ONLY),
                       ("IPP_ADAPTIVE", _ffi.IPP_ADAPTIVE), ("IPP_USE_FLIGHT_TIME", _ffi.IPP_USE_FLIGHT_TIME),
                       ("IPP_GIVEN_OBSERVATION", _ffi.IPP_GIVEN_OBSERVATION), ("IPP_MAX_MEAS", _ffi.IPP_MAX_MEAS)):
        m = re.search(rf"#define\s+{macro}\s+(\d+)", txt)
        assert m and int(m.group(1)) == val


def test_struct_layouts_match_header_order():
    from ipp_rl_amd import _ffi

    txt = open(os.path.join(ROOT, "include", "ipp_engine.h")).read()
    body = re.search(r"typedef struct ipp_config \{(.*?)\} ipp_config;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:int32_t|double)\s+([a-z_0-9]+);", body)
    assert fields == [f[0] for f in _ffi.IppConfig._fields_]
    assert ctypes.sizeof(_ffi.IppConfig) == 8 + 13 * 8 + 10 * 4
    # ipp_mcts_tables: same field order; int32 / double scalars, everything else a device pointer
    body = re.search(r"typedef struct ipp_mcts_tables \{(.*?)\} ipp_mcts_tables;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split(";"):
        m = re.match(r"\s*(?:const\s+)?(int32_t|int64_t|double|uint64_t|uint8_t|float)\s*(\*?)\s*(.*)", stmt, flags=re.S)
        if not m or not m.group(3).strip():
            continue
        for name in m.group(3).split(","):
            is_ptr = bool(m.group(2)) or name.strip().startswith("*")
            fields.append((name.strip().lstrip("*").strip(), ctypes.c_void_p if is_ptr else {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "double": ctypes.c_double}[m.group(1)]))
    assert fields == list(_ffi.IppMctsTables._fields_)
    # (.. + the optional Ns tables and their length + the four int32 of a search split into groups of roots)
    assert ctypes.sizeof(_ffi.IppMctsTables) == 16 * 4 + 8 * 8 + 46 * 8 + 2 * 8 + 8 + 4 * 4


def test_arena_sizing_and_validation_without_gpu():
    """Pure host entry points work without a device; bad configs are rejected with a message."""
    from ipp_rl_amd import _ffi

    lib = _ffi.load()
    c = _ffi.IppConfig(x_dim=50, y_dim=50, resolution=4.0, tan_half_fov_x=0.57735, tan_half_fov_y=0.57735,
                       rf_altitude=10.0, coeff_a=0.05, coeff_b=0.2, signal_variance=1.82, length_scale=3.67, max_v=2,
                       max_a=2, value_threshold=0.4, interval_factor=0, cluster_radius=5, state_repr=_ffi.IPP_FACTOR,
                       capacity=4096, rank_cap=360, max_batch=4096, max_measurements=9, tile_threads=0)
    nbytes = ctypes.c_uint64(0)
    assert lib.ipp_engine_arena_bytes(ctypes.byref(c), ctypes.byref(nbytes)) == 0
    cov = 4096 * 360 * 2560 * 4
    assert cov < nbytes.value < cov * 1.05  # state is dominated by the factor slab U[B][r_cap][Npad]
    c.state_repr = _ffi.IPP_DENSE
    assert lib.ipp_engine_arena_bytes(ctypes.byref(c), ctypes.byref(nbytes)) == 0
    assert nbytes.value > 4096 * 2500 * 2560 * 4
    c.rank_cap, c.state_repr = 0, _ffi.IPP_FACTOR
    assert lib.ipp_engine_arena_bytes(ctypes.byref(c), ctypes.byref(nbytes)) < 0
    assert b"rank_cap" in lib.ipp_last_error()
    c.rank_cap, c.tile_threads = 360, 100
    assert lib.ipp_engine_arena_bytes(ctypes.byref(c), ctypes.byref(nbytes)) < 0


def test_arena_calls_validate_their_arguments_without_gpu():
    """ipp_arena_* (include/ipp_engine.h "Arena placement"): argument errors are reported before any device call; freeing nothing is fine; the
    Python layer refuses to build a DeviceArena without a device (no host fallback for device memory)."""
    from ipp_rl_amd import _ffi, engine

    lib = _ffi.load()
    out = ctypes.c_void_p()
    assert lib.ipp_arena_alloc(0, 0, _ffi.IPP_ARENA_VMM, 0, 0, ctypes.byref(out)) == -1 and b"ipp_arena_alloc" in lib.ipp_last_error()
    assert lib.ipp_arena_alloc(0, 1 << 20, _ffi.IPP_ARENA_VMM, 0, 0, None) == -1
    assert lib.ipp_arena_free(None) == 0
    assert lib.ipp_arena_free(ctypes.c_void_p(0x1000)) == -1 and b"not an arena" in lib.ipp_last_error()
    ms, n = ctypes.c_double(), ctypes.c_uint64(123)
    assert lib.ipp_arena_probe(0, None, 1 << 20, 4, 4, 1, None, ctypes.byref(ms)) == -1
    assert lib.ipp_arena_probe(0, ctypes.c_void_p(0x1000), 1 << 20, 0, 4, 1, None, ctypes.byref(ms)) == -1
    assert lib.ipp_arena_probe(0, ctypes.c_void_p(0x1000), 64, 4, 4, 1, None, ctypes.byref(ms)) == -1 and b"smaller than one patch" in lib.ipp_last_error()
    assert lib.ipp_arena_latency(0, None, 1 << 20, 4, 4, None, ctypes.byref(ms)) == -1
    assert lib.ipp_arena_retired_bytes(None) == -1
    assert lib.ipp_arena_retired_bytes(ctypes.byref(n)) == 0 and n.value == 0
    n.value = 123
    assert lib.ipp_arena_trim(-1, ctypes.byref(n)) == 0 and n.value == 0 and lib.ipp_arena_trim(0, None) == 0  # (nothing pooled)
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(_ffi.IppError):
            engine.DeviceArena(1 << 20, 0, kind="vmm")
    assert engine.ARENA_VMM_MIN_BYTES == 256 << 20


def test_tree_patches_stay_within_reach_of_their_32_bit_record_offsets():
    """k_tree_patch addresses a column patch by a 32-bit offset in 8-byte units from View::cov: root slots and node blocks have to lie
    within 2^35 bytes of it.  The decision is taken on the finished arena layout (the score scratch -- one dense P, 17 GB at
    256x256 -- sits between the two regions: ADVICE r03); beyond the reach the engine plans the band-tile tree kernels, whose nodes
    are an order of magnitude larger."""
    from ipp_rl_amd import _ffi

    lib = _ffi.load()

    def arena(nodes, score):
        c = _ffi.IppConfig(x_dim=256, y_dim=256, resolution=4.0, tan_half_fov_x=0.57735, tan_half_fov_y=0.57735,
                           rf_altitude=10.0, coeff_a=0.05, coeff_b=0.2, signal_variance=1.82, length_scale=3.67, max_v=2,
                           max_a=2, value_threshold=0.4, interval_factor=0, cluster_radius=5, state_repr=_ffi.IPP_FACTOR,
                           capacity=4, rank_cap=90, max_batch=64, max_measurements=9, tile_threads=0, window_rows=10,
                           score_scratch=score, node_capacity=nodes, fixed_prior=1)
        n = ctypes.c_uint64(0)
        assert lib.ipp_engine_arena_bytes(ctypes.byref(c), ctypes.byref(n)) == 0, lib.ipp_last_error()
        return n.value

    per_node_patch = (arena(200_000, 0) - arena(100_000, 0)) / 100_000
    assert 20e3 < per_node_patch < 40e3                      # a node = 10 patches of ~2.6 KB
    # without the scratch 700 000 nodes are 19 GB: in reach; with the 17 GB scratch in front of them they are not
    assert arena(700_000, 0) < 22e9
    assert arena(700_000, 1) - arena(700_000, 0) > 100e9     # band-tile nodes (>= 25 grid rows of 256 cells x 10 per node)
    assert arena(100_000, 1) - arena(100_000, 0) < 20e9      # 2.7 GB of nodes behind the scratch: still patches


def test_engine_fails_loudly_without_gpu():
    import torch

    from ipp_rl_amd import EngineConfig, IPPEngine, IppError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(IppError):
        IPPEngine(EngineConfig(x_dim=10, y_dim=10), capacity=2)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from ipp_rl_amd import _ffi

    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.IppError, match="no CPU fallback"):
        _ffi.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ipp-rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "libipp_oracle" not in src


# ------------------------------------------------------------------ host mirrors vs golden
def make_sensor(dim_x, dim_y, res):
    from ipp_rl_amd.mapping.grid_maps import GridMap
    from ipp_rl_amd.sensors.cameras import RGBCamera
    from ipp_rl_amd.sensors.models.sensor_models import AltitudeSensorModel

    params = example_params(dim_x, dim_y, res)
    gm = GridMap(params)
    return gm, RGBCamera(params["sensor"]["field_of_view"], AltitudeSensorModel(0.05, 0.2), gm)


def test_camera_footprints_vs_golden(golden):
    g = golden("footprints")
    for res, alt, rx, ry, radx, rady, rf, nv in g["table"]:
        gm, cam = make_sensor(400, 400, res)
        pos = np.array([200.0 * res + 0.5 * res, 200.0 * res + 0.5 * res, alt])
        assert cam.field_of_view_range(alt) == (rx, ry)
        xl, xr, yu, yd = cam.project_field_of_view(pos)
        assert ((xr - xl) // 2, (yd - yu) // 2) == (radx, rady)
        assert cam.get_resolution_factor(pos) == rf and cam.sensor_model.get_noise_variance(pos) == nv
    gm, cam = make_sensor(50, 50, 4)
    for p, fov in zip(g["positions"], g["fovs"]):
        assert cam.project_field_of_view(p) == tuple(fov)


def test_sensor_model_matrices_vs_golden(golden):
    g = golden("measurement_model")
    gm, cam = make_sensor(int(g["x_dim"]), int(g["y_dim"]), 4)
    sm = cam.sensor_model
    for fov, rf, m, H, r00 in zip(g["fovs"], g["rfs"], g["ms"], g["H"], g["R00"]):
        Hm = sm.measurement_model_matrix(gm, tuple(int(v) for v in fov), int(m), int(rf))
        assert np.array_equal(Hm, H[: int(m)])
        R = sm.measurement_variance_matrix(np.array([0, 0, 14.0 if rf == 2 else 8.0]), int(m), int(rf))
        assert abs(R[0, 0] - r00) < 1e-16 and R.shape == (m, m)


def test_action_costs_and_rewards_vs_golden(golden):
    from ipp_rl_amd.planning.common import actions, rewards

    g = golden("costs")
    uav, uav2 = {"max_v": 2, "max_a": 2}, {"max_v": 5.0, "max_a": 1.5}
    for i, (a, b) in enumerate(zip(g["a"], g["b"])):
        assert abs(actions.action_costs(a, b, None) - g["dist"][i]) < 1e-12
        assert abs(actions.action_costs(a, b, uav) - g["t_v2a2"][i]) < 1e-12
        assert abs(actions.compute_flight_time(a, b, uav2) - g["t_v5a15"][i]) < 1e-12
    assert np.max(np.abs(actions.compute_flight_times(g["a"], g["b"][0], uav) - g["t_vec"])) < 1e-12
    gp = golden("predict_10")
    P0, P1 = gp["P0"], gp["P_seq"][0]
    a0, prev = gp["actions"][0], np.array([2.0, 2.0, 14.0])
    msk = rewards.compute_adaptive_msk(gp["mean_used"][0], P0, 0.4, 0)
    assert np.array_equal(msk, gp["mask"][0])
    assert abs(rewards.compute_reward(P0, P1, prev, a0, uav, msk) - gp["reward"][0]) < 1e-12
    gg = golden("greedy")
    gm, _ = make_sensor(10, 10, 4)
    cands = actions.get_actions(prev, 200, gm, 8, 14, 6, uav)
    assert np.array_equal(np.array(cands), gg["candidates_10"])
    en = actions.enumerate_actions(gm, 8, 14, 6)
    assert len(en) == 200 and np.array_equal(en[0], [2.0, 2.0, 8.0]) and np.array_equal(en[100 + 10], [6.0, 2.0, 14.0])
    assert actions.out_of_bounds([41, 2, 8], gm, 8, 14) and not actions.out_of_bounds([40, 2, 8], gm, 8, 14)


def test_config_errors_follow_the_reference():
    from ipp_rl_amd.mapping.grid_maps import GridMap

    with pytest.raises(ValueError):
        GridMap({"environment": {"x_dim": 3}}).y_dim
    with pytest.raises(ValueError):
        GridMap({}).x_dim


def test_engine_config_from_params_and_workload_helpers():
    from ipp_rl_amd import EngineConfig
    from ipp_rl_amd.vec_env import cell_centre_actions, shard_range

    cfg = EngineConfig.from_params(example_params(50))
    assert (cfg.x_dim, cfg.resolution, cfg.coeff_b, cfg.signal_variance, cfg.max_v) == (50, 4.0, 0.2, 1.82, 2.0)
    full = cell_centre_actions(cfg, 3, 0, 64, 64, [5.0, 6.0, 7.0])
    parts = [cell_centre_actions(cfg, 3, *shard_range(64, r, 4), 64, [5.0, 6.0, 7.0]) for r in range(4)]
    assert np.array_equal(np.concatenate(parts), full)  # sharding does not change any env's inputs
    assert np.all((full[:, 0] - 2.0) % 4.0 == 0) and set(np.unique(full[:, 2])) <= {5.0, 6.0, 7.0}
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]


def test_forked_child_abandons_inherited_engines_and_refuses_new_ones():
    """HIP state does not survive fork(): the at-fork hook drops every live engine's handle WITHOUT calling
    ipp_engine_destroy (no GPU needed to check that: a stand-in object is registered as a live engine), and both
    IPPEngine() and the compat runtime raise an IppError that names the spawn start method."""
    import os

    from ipp_rl_amd import EngineConfig, _runtime, engine
    from ipp_rl_amd._ffi import IppError

    class Standin:
        def __init__(self):
            self._h, self.arena, self.destroyed = 123, object(), False

    live = Standin()
    engine._LIVE.add(live)
    assert not engine.forked_with_gpu()
    pid = os.fork()
    if pid == 0:
        code = 0
        try:
            ok = live._h is None and engine.forked_with_gpu() and any(a is live.arena for _, a in engine._ABANDONED)
            for make in (lambda: engine.IPPEngine(EngineConfig(), capacity=1), lambda: _runtime.compat_engine(EngineConfig())):
                try:
                    make()
                    ok = False
                except IppError as exc:
                    ok = ok and "spawn" in str(exc)
            code = 0 if ok else 1
        except BaseException:  # noqa: BLE001
            code = 2
        os._exit(code)
    assert os.waitpid(pid, 0)[1] == 0
    assert live._h == 123 and not engine.forked_with_gpu()  # the parent is unaffected
    engine._LIVE.discard(live)


def test_device_cov_semantics_without_a_gpu():
    """DeviceCov / SlotStore (ipp_rl_amd/_device_array.py) against a stand-in engine: diag / trace from the 'device',
    arithmetic and indexing materialise once, LRU eviction to the host, write-through detaches and re-uploads, pickling
    and deepcopy give plain ndarrays, garbage collection returns the slot."""
    import copy
    import gc
    import pickle

    from ipp_rl_amd._device_array import DeviceCov, SlotStore

    class T:
        def __init__(self, a): self.a = a
        def detach(self): return self
        def cpu(self): return self
        def numpy(self): return self.a
        def double(self): return T(self.a.astype(np.float64))
        def numel(self): return self.a.size

    class FakeEngine:
        def __init__(self): self.slots = {}
        def read_cov(self, slot): return T(self.slots[slot].astype(np.float32))
        def read_diag(self, slot): return T(np.diag(self.slots[slot]).astype(np.float32))
        def write_cov(self, slot, P): self.slots[slot] = np.array(P, dtype=np.float64)

    eng = FakeEngine()
    store = SlotStore(eng, 2, 3)
    A = np.arange(16.0).reshape(4, 4)
    a = DeviceCov(store, 4, host=A.copy())
    assert a.device_slot(store) == 2 and store.uploads == 1
    b = DeviceCov.new_on_device(store, 4)
    eng.slots[b._slot] = A * 2
    assert np.array_equal(np.diag(b), np.diag(A * 2)) and np.trace(b) == 60.0 and store.downloads == 0
    assert np.array_equal(b - a, A) and np.array_equal(np.asarray(b), A * 2) and b.shape == (4, 4) and len(b) == 4
    assert store.downloads == 1 and np.array_equal(b[1], 2 * A[1]) and store.downloads == 1  # materialised once
    c = DeviceCov.new_on_device(store, 4)
    eng.slots[c._slot] = A * 3
    store.touch(b._slot)
    d = DeviceCov.new_on_device(store, 4)  # no free slot: the least recently used state (a) is parked on the host
    assert store.evictions == 1 and a._slot is None and np.array_equal(np.asarray(a), A)
    eng.slots[d._slot] = A * 4
    c[0, :] = 0
    assert c._slot is None and np.asarray(c)[0].sum() == 0 and np.asarray(c)[1, 1] == 15
    s2 = c.device_slot(store)
    assert eng.slots[s2][0].sum() == 0 and eng.slots[s2][1, 1] == 15  # the modified matrix went back up
    assert isinstance(pickle.loads(pickle.dumps(d)), np.ndarray) and isinstance(copy.deepcopy(d), np.ndarray)
    n_free = len(store.free)
    del d
    gc.collect()
    assert len(store.free) == n_free + 1
    assert DeviceCov(SlotStore(FakeEngine(), 0, 1), 4, host=A).device_slot(store) is None  # another engine's state
