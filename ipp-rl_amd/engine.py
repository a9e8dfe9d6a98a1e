"""
Python handle on the HIP step engine (C-ABI in include/ipp_engine.h).

PyTorch-ROCm is used only as the device-memory container (one uint8 arena tensor + I/O tensors whose
``data_ptr()`` crosses the ABI) and for the current HIP stream; every computation happens in the HIP
kernels of ``csrc/``.  There is no CPU path: constructing an engine without the library or without a
GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from . import _ffi


@dataclass
class EngineConfig:
    """Hot-path parameters (the YAML keys of SURVEY section 5; reference config/example.yaml)."""

    x_dim: int = 50
    y_dim: int = 50
    resolution: float = 4.0
    angle_x: float = 60.0
    angle_y: float = 60.0
    rf_altitude: float = 10.0  # sensors/cameras.py:125
    coeff_a: float = 0.05
    coeff_b: float = 0.2
    signal_variance: float = 1.82
    length_scale: float = 3.67
    max_v: float = 2.0
    max_a: float = 2.0
    value_threshold: float = 0.4
    interval_factor: float = 0.0
    cluster_radius: float = 5.0

    @property
    def n_cells(self) -> int:
        return self.x_dim * self.y_dim

    @classmethod
    def from_params(cls, params: Dict) -> "EngineConfig":
        env, sen, mp = params["environment"], params["sensor"], params["mapping"]
        exp = params.get("experiment", {})
        uav, scen = exp.get("uav", {}), exp.get("scenario", {})
        return cls(
            x_dim=int(env["x_dim"]), y_dim=int(env["y_dim"]), resolution=float(env["resolution"]),
            angle_x=float(sen["field_of_view"]["angle_x"]), angle_y=float(sen["field_of_view"]["angle_y"]),
            coeff_a=float(sen["model"]["coeff_a"]), coeff_b=float(sen["model"]["coeff_b"]),
            signal_variance=float(mp["signal_variance"]), length_scale=float(mp["length_scale"]),
            max_v=float(uav.get("max_v", 2.0)), max_a=float(uav.get("max_a", 2.0)),
            value_threshold=float(scen.get("value_threshold", 0.4)),
            interval_factor=float(scen.get("interval_factor", 0.0)),
            cluster_radius=float(sen.get("simulation", {}).get("cluster_radius", 5.0)),
        )


def _torch():
    import torch

    return torch


# ---- fork safety.  HIP state (streams, events, device memory) does not survive fork(), and PyTorch refuses to
# re-initialise the device in a forked child.  A child forked from a process with live engines therefore ABANDONS them:
# the handle is dropped without ipp_engine_destroy (hipStreamDestroy on the parent's objects can hang) and the arena
# tensors are kept referenced forever so that nothing tries to free them.  New engines cannot be created there.
FORK_MESSAGE = ("this process was fork()ed from a parent that had already opened the GPU engine; HIP state does not survive "
                "fork.  Start workers with the spawn method (multiprocessing.set_start_method('spawn'), "
                "multiprocessing.get_context('spawn').Pool(...), torch.multiprocessing.spawn): a spawned worker opens its own "
                "engine.")
ARENA_VMM_MIN_BYTES = 256 << 20  # arena="auto": arenas of at least this size come from the virtual-memory API (DeviceArena "vmm")
_LIVE = weakref.WeakSet()
_ABANDONED = []
_forked_with_gpu = False


def forked_with_gpu() -> bool:
    return _forked_with_gpu


def _after_fork_in_child():
    global _forked_with_gpu
    live = list(_LIVE)
    if not live:
        return
    for eng in live:
        _ABANDONED.append((eng, getattr(eng, "arena", None)))
        eng._h = None
    _forked_with_gpu = True


os.register_at_fork(after_in_child=_after_fork_in_child)


class DeviceArena:
    """Device memory for an engine's state straight from the driver (ipp_arena_alloc): kind "hip" = hipMalloc, "vmm" = a
    1-GiB-aligned reservation backed by physical chunks of `chunk_bytes` (0 = 1 GiB).  No caching allocator in between, so
    freeing it returns the physical memory (a "vmm" arena's ADDRESS range stays reserved, ipp_arena_free); probe() / latency() time the
    bare row stream and chains of dependent requests on it (profiles/r06_arena_modes.txt)."""

    def __init__(self, nbytes: int, device_index: int = 0, kind: str = "hip", chunk_bytes: int = 0, align_bytes: int = 0):
        self._lib = _ffi.load()
        if not _torch().cuda.is_available():
            raise _ffi.IppError("DeviceArena needs a HIP device")
        _torch().cuda.init()
        kinds = {"hip": _ffi.IPP_ARENA_HIPMALLOC, "vmm": _ffi.IPP_ARENA_VMM}
        if kind not in kinds:
            raise ValueError("kind must be 'hip' or 'vmm'")
        self.kind, self.nbytes, self.device_index = kind, int(nbytes), int(device_index)
        p = C.c_void_p()
        _ffi.check(self._lib.ipp_arena_alloc(self.device_index, self.nbytes, kinds[kind], int(chunk_bytes), int(align_bytes), C.byref(p)))
        self._ptr = p.value
        self._pid = os.getpid()
        self.chunk_bytes = int(chunk_bytes) if kind == "vmm" else 0
        if os.environ.get("IPP_ARENA_LOG"):
            import sys
            print(f"[arena] alloc {kind} 0x{self._ptr:x} .. 0x{self._ptr + self.nbytes:x} ({self.nbytes} bytes, chunk {self.chunk_bytes >> 20} MiB)", file=sys.stderr, flush=True)

    def data_ptr(self) -> int:
        return self._ptr

    def probe(self, items: int, rows: int = 32, launches: int = 5, nbytes: Optional[int] = None) -> float:
        """ms per launch of the bare row stream of the step kernel over this arena (ipp_arena_probe)."""
        ms = C.c_double(0.0)
        stream = C.c_void_p(_torch().cuda.current_stream(self.device_index).cuda_stream)
        _ffi.check(self._lib.ipp_arena_probe(self.device_index, C.c_void_p(self._ptr), int(nbytes or self.nbytes), int(items), int(rows),
                                             int(launches), stream, C.byref(ms)))
        return float(ms.value)

    def latency(self, waves: int = 256, hops: int = 2000, nbytes: Optional[int] = None) -> float:
        """ns per hop of chains of dependent requests to random patches of this arena (ipp_arena_latency)."""
        ns = C.c_double(0.0)
        stream = C.c_void_p(_torch().cuda.current_stream(self.device_index).cuda_stream)
        _ffi.check(self._lib.ipp_arena_latency(self.device_index, C.c_void_p(self._ptr), int(nbytes or self.nbytes), int(waves), int(hops),
                                               stream, C.byref(ns)))
        return float(ns.value)

    def as_tensor(self, device):
        """A uint8 view of the arena (tests: poisoning); valid while the arena lives."""
        torch = _torch()

        class _Iface:
            pass

        h = _Iface()
        h.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self._ptr, False), "version": 2}
        return torch.as_tensor(h, device=device)

    @staticmethod
    def trim(device_index: int = -1) -> int:
        """Hands the pooled physical chunks of freed "vmm" arenas back to the driver (ipp_arena_trim); returns the bytes released."""
        n = C.c_uint64(0)
        _ffi.check(_ffi.load().ipp_arena_trim(int(device_index), C.byref(n)))
        return int(n.value)

    def free(self):
        if getattr(self, "_ptr", None) and os.getpid() != self._pid:
            self._ptr = None  # a forked child: the mapping is the parent's (HIP state does not survive fork)
        if getattr(self, "_ptr", None):
            if os.environ.get("IPP_ARENA_LOG"):
                import sys
                print(f"[arena] free 0x{self._ptr:x}", file=sys.stderr, flush=True)
            _torch().cuda.synchronize(self.device_index)
            _ffi.check(self._lib.ipp_arena_free(C.c_void_p(self._ptr)))
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class IPPEngine:
    """B environment slots on one GPU; state lives in one caller-owned arena tensor."""

    def __init__(self, cfg: EngineConfig, capacity: int, state: str = "factor", rank_cap: int = 360,
                 max_batch: Optional[int] = None, device: str = "cuda:0", max_measurements: int = 9,
                 tile_threads: int = 0, window_rows: int = 0, score_scratch: bool = False, node_capacity: int = 0,
                 fixed_prior: bool = False, arena=None):
        """arena: None / "auto" = a torch.uint8 tensor for small arenas, a "vmm" DeviceArena from 256 MiB; "torch"; "hip" / "vmm" = a
        DeviceArena of that kind owned by the engine; a DeviceArena instance = the caller's (the engine never frees it).
        window_rows: 0 = exact columns, R > 0 = columns kept within R grid rows of their footprint, -1 = the smallest
        R the engine accepts for this prior (ipp_min_window_rows).  fixed_prior: no reset will install a length scale above
        cfg.length_scale (no shuffle_prior_cov), which lets the window be 10 instead of 12 rows for the example config."""
        if _forked_with_gpu:
            raise _ffi.IppError(FORK_MESSAGE)
        torch = _torch()
        self._lib = _ffi.load()
        if not torch.cuda.is_available():
            raise _ffi.IppError("IPPEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.capacity = int(capacity)
        self.state = state
        self.max_batch = int(max_batch or capacity)
        c = _ffi.IppConfig()
        c.x_dim, c.y_dim, c.resolution = cfg.x_dim, cfg.y_dim, cfg.resolution
        # tan(0.5 * radians(angle)) in fp64 on the host, exactly as sensors/cameras.py:44-45 evaluates it
        c.tan_half_fov_x = float(np.tan(0.5 * np.radians(cfg.angle_x)))
        c.tan_half_fov_y = float(np.tan(0.5 * np.radians(cfg.angle_y)))
        c.rf_altitude = cfg.rf_altitude
        c.coeff_a, c.coeff_b = cfg.coeff_a, cfg.coeff_b
        c.signal_variance, c.length_scale = cfg.signal_variance, cfg.length_scale
        c.max_v, c.max_a = cfg.max_v, cfg.max_a
        c.value_threshold, c.interval_factor = cfg.value_threshold, cfg.interval_factor
        c.cluster_radius = cfg.cluster_radius
        c.state_repr = _ffi.IPP_FACTOR if state == "factor" else _ffi.IPP_DENSE
        if state not in ("factor", "dense"):
            raise ValueError("state must be 'factor' or 'dense'")
        c.capacity, c.rank_cap, c.max_batch = self.capacity, int(rank_cap), self.max_batch
        c.max_measurements, c.tile_threads = int(max_measurements), int(tile_threads)
        c.fixed_prior = 1 if fixed_prior else 0
        c.window_rows = int(window_rows)
        if c.window_rows < 0:
            if state != "factor":
                c.window_rows = 0
            else:
                rows = C.c_int32(0)
                _ffi.check(self._lib.ipp_min_window_rows(C.byref(c), C.byref(rows)))
                c.window_rows = int(rows.value)
        c.score_scratch = 1 if score_scratch else 0
        c.node_capacity = int(node_capacity)
        self._c = c
        nbytes = C.c_uint64(0)
        _ffi.check(self._lib.ipp_engine_arena_bytes(C.byref(c), C.byref(nbytes)))
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        # the arena: a torch tensor (default), or memory straight from the driver (DeviceArena: "hip" = hipMalloc, "vmm" = the
        # virtual-memory API) -- an arena the caching allocator has never seen, whose physical placement the caller can re-draw
        self.arena_bytes = int(nbytes.value) + 256
        if arena is None:
            arena = os.environ.get("IPP_ARENA", "auto")
        self._own_arena = isinstance(arena, str) and arena != "torch"
        self.arena_kind = arena if isinstance(arena, str) else "caller"
        if isinstance(arena, str) and arena not in ("torch", "auto", "hip", "vmm"):
            raise ValueError("arena must be 'auto', 'torch', 'hip', 'vmm' or a DeviceArena")
        if arena == "auto":
            # Large arenas from the virtual-memory API in 1-GiB physical chunks at 1-GiB-aligned addresses: the driver then maps
            # them with large translation fragments, and a batch whose items walk a few hundred MB .. tens of GB of state runs
            # in what rounds 4-5 knew as the "fast mode" in EVERY process (configs[3] share 0.597 -> 0.511 ms per step, configs[2]
            # 0.592 -> 0.488; torch.empty / hipMalloc land there by chance: profiles/r06_arena_modes.txt).  Small arenas stay
            # torch tensors (tests build hundreds of engines; a mapping costs milliseconds and rounds up to the granularity).
            arena = "torch"
            if self.arena_bytes >= ARENA_VMM_MIN_BYTES:
                try:
                    chunk = int(os.environ.get("IPP_ARENA_CHUNK_MIB", "1024")) << 20
                    while chunk > (64 << 20) and chunk > self.arena_bytes:
                        chunk >>= 1
                    arena = DeviceArena(self.arena_bytes, dev_index, kind="vmm", chunk_bytes=chunk, align_bytes=chunk)
                    self.arena_kind = "vmm"
                except _ffi.IppError as e:  # (no VMM support / no contiguous physical memory left: the allocator's memory then)
                    self.arena_fallback_reason = str(e)
                    self.arena_kind = "torch"
            else:
                self.arena_kind = "torch"
        elif isinstance(arena, str) and arena != "torch":
            chunk = int(os.environ.get("IPP_ARENA_CHUNK_MIB", "1024")) << 20  # (A/B; aligned to its own size)
            arena = DeviceArena(self.arena_bytes, dev_index, kind=arena, chunk_bytes=chunk, align_bytes=chunk)
        if isinstance(arena, DeviceArena):
            if arena.nbytes < self.arena_bytes:
                raise ValueError(f"arena of {arena.nbytes} bytes for an engine that needs {self.arena_bytes}")
            self.arena = arena
        else:
            self.arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.device)
        if os.environ.get("IPP_POISON_ARENA"):  # tests: every float the engine does not initialise reads as NaN
            if isinstance(self.arena, DeviceArena):
                self.arena.as_tensor(self.device).fill_(0xFF)
            else:
                self.arena.fill_(0xFF)
        base = (self.arena.data_ptr() + 255) // 256 * 256
        handle = C.c_void_p()
        _ffi.check(self._lib.ipp_engine_create(C.byref(c), dev_index, C.c_void_p(base), nbytes.value, C.byref(handle)))
        self._h = handle
        _LIVE.add(self)
        info = _ffi.IppInfo()
        _ffi.check(self._lib.ipp_engine_info(self._h, C.byref(info)))
        self.info = info
        self.n_cells, self.n_pad, self.meas_cap = info.n_cells, info.n_pad, info.meas_cap
        self.rank_cap = int(rank_cap)

    # ------------------------------------------------------------------ plumbing
    def close(self):
        """Destroys the engine and releases the arena -- after EVERYTHING issued on the device has finished: launches of
        step_parts / VecIPPEnv.step_async run on streams the caching allocator does not associate with the arena tensor."""
        if getattr(self, "_h", None):
            try:
                _torch().cuda.synchronize(self.device)
            except Exception:
                pass
            self._lib.ipp_engine_destroy(self._h)
            self._h = None
            if getattr(self, "_own_arena", False) and isinstance(self.arena, DeviceArena):
                self.arena.free()
            self.arena = None
            self._parts_ring = None
            self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype, shape=None):
        """Device tensor of `dtype` (contiguous) from a tensor / ndarray / list; None passes through."""
        torch = _torch()
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=dtype).contiguous()
        else:
            arr = np.ascontiguousarray(x)
            if arr.size >= (1 << 16) and arr.dtype == np.float64 and dtype == torch.float32:
                # large fp64 host arrays (dense covariances of the drop-in classes): copy as they are and convert on
                # the device -- numpy's single-threaded astype costs more than the extra PCIe bytes
                t = torch.from_numpy(arr).to(self.device).to(dtype)
            else:
                t = torch.as_tensor(arr, dtype=dtype, device=self.device)
        if shape is not None:
            t = t.reshape(shape)
        return t

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    # ------------------------------------------------------------------ C-ABI calls
    def reset(self, env_ids=None, prior_scale=None, gt=None, white_noise=None, n: Optional[int] = None, prev=None,
              init_action=None):
        """prev / init_action: also return the UAVs of the reset envs to the mission start in the same kernel
        (prev: [capacity, 3] float64 device tensor indexed by env id; ipp_reset_episode)."""
        torch = _torch()
        ids = self._dev(env_ids, torch.int32)
        n = int(ids.numel()) if ids is not None else int(n if n is not None else self.capacity)
        ps = self._dev(prior_scale, torch.float64, (n, 2))
        g = self._dev(gt, torch.float32, (n, self.n_cells))
        w = self._dev(white_noise, torch.float32, (n, self.n_cells))
        if prev is None:
            _ffi.check(self._lib.ipp_reset(self._h, self._ptr(ids), n, self._ptr(ps), self._ptr(g), self._ptr(w), self.stream))
        else:
            if prev.dtype != torch.float64 or not prev.is_contiguous() or prev.numel() != 3 * self.capacity:
                raise ValueError("prev must be a contiguous float64 [capacity, 3] device tensor")
            ia = (C.c_double * 3)(*[float(x) for x in init_action])
            _ffi.check(self._lib.ipp_reset_episode(self._h, self._ptr(ids), n, self._ptr(ps), self._ptr(g), self._ptr(w),
                                                   self._ptr(prev), ia, self.stream))
        self._keep = (ids, ps, g, w)

    def score_actions(self, env: int, actions, prev_action, *, adaptive=True, use_flight_time=True, reward_out=None,
                      status_out=None):
        """Reward of every candidate action from the current state of slot `env` (nothing is written): the state
        is read once for all candidates (ipp_score_actions).  The engine needs score_scratch=True."""
        torch = _torch()
        a = self._dev(actions, torch.float64).reshape(-1, 3)
        n = a.shape[0]
        reward = reward_out if reward_out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        status = status_out if status_out is not None else torch.empty(n, dtype=torch.int32, device=self.device)
        pv = (C.c_double * 3)(*[float(x) for x in np.asarray(prev_action, dtype=np.float64).ravel()[:3]])
        flags = (_ffi.IPP_ADAPTIVE if adaptive else 0) | (_ffi.IPP_USE_FLIGHT_TIME if use_flight_time else 0)
        _ffi.check(self._lib.ipp_score_actions(self._h, int(env), self._ptr(a), n, pv, flags, self._ptr(reward),
                                               self._ptr(status), self.stream))
        self._keep_score = a
        return reward, status

    TREE_DEPTH = 6

    def tree_step(self, root_ids, path_ids, actions, prev_actions, new_ids=None, *, adaptive=True, use_flight_time=True,
                  reward_out=None, status_out=None):
        """Covariance-only predict steps at tree nodes (ipp_tree_step): item i starts from env slot root_ids[i] plus
        the nodes path_ids[i] (root side first, -1 padded to TREE_DEPTH); new_ids[i] >= 0 records the step as that node."""
        torch = _torch()
        roots = self._dev(root_ids, torch.int32)
        n = int(roots.numel())
        paths = self._dev(path_ids, torch.int32).reshape(n, self.TREE_DEPTH)
        new = self._dev(new_ids, torch.int32)
        a = self._dev(actions, torch.float64).reshape(n, 3)
        p = self._dev(prev_actions, torch.float64).reshape(n, 3)
        reward = reward_out if reward_out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        status = status_out if status_out is not None else torch.empty(n, dtype=torch.int32, device=self.device)
        flags = (_ffi.IPP_ADAPTIVE if adaptive else 0) | (_ffi.IPP_USE_FLIGHT_TIME if use_flight_time else 0) | \
                (_ffi.IPP_PREDICT_ONLY if new is None else 0)
        _ffi.check(self._lib.ipp_tree_step(self._h, self._ptr(roots), self._ptr(paths), self._ptr(new), n, self._ptr(a),
                                           self._ptr(p), flags, self._ptr(reward), self._ptr(status), self.stream))
        self._keep = (roots, paths, new, a, p)
        return reward, status

    def tree_score_actions(self, root: int, path, actions, prev_action, *, adaptive=True, use_flight_time=True):
        """score_actions from the state of a tree node (root env slot + node path, host list padded with -1)."""
        torch = _torch()
        a = self._dev(actions, torch.float64).reshape(-1, 3)
        n = a.shape[0]
        reward = torch.empty(n, dtype=torch.float32, device=self.device)
        status = torch.empty(n, dtype=torch.int32, device=self.device)
        ids = (C.c_int32 * self.TREE_DEPTH)(*[int(x) for x in (list(path) + [-1] * self.TREE_DEPTH)[: self.TREE_DEPTH]])
        pv = (C.c_double * 3)(*[float(x) for x in np.asarray(prev_action, dtype=np.float64).ravel()[:3]])
        flags = (_ffi.IPP_ADAPTIVE if adaptive else 0) | (_ffi.IPP_USE_FLIGHT_TIME if use_flight_time else 0)
        _ffi.check(self._lib.ipp_tree_score_actions(self._h, int(root), ids, self._ptr(a), n, pv, flags, self._ptr(reward),
                                                    self._ptr(status), self.stream))
        self._keep_score = a
        return reward, status

    def tree_diag(self, node: int):
        torch = _torch()
        out = torch.empty(self.n_cells, dtype=torch.float32, device=self.device)
        _ffi.check(self._lib.ipp_tree_read_diag(self._h, int(node), self._ptr(out), self.stream))
        return out

    def state_plane(self, env: int, mean_for_mask=None, adaptive: bool = True, out=None):
        """Masked, min-max normalised N x N covariance plane of slot `env` (features.py:91-101), device fp32 [N, N].
        mean_for_mask: the map mean the mask is taken from (None: the slot's own).  Needs score_scratch=True."""
        torch = _torch()
        mu = self._dev(mean_for_mask, torch.float32)
        if mu is not None:
            mu = mu.reshape(-1)
        if out is None:
            out = torch.empty((self.n_cells, self.n_cells), dtype=torch.float32, device=self.device)
        _ffi.check(self._lib.ipp_state_plane(self._h, int(env), self._ptr(mu), _ffi.IPP_ADAPTIVE if adaptive else 0,
                                             self._ptr(out), self.stream))
        self._keep_plane = mu
        return out

    def generate_grf(self, white_noise, out=None, stream=None):
        """white noise [n, N] -> normalised GRF [n, N] in a caller tensor (no env slot touched)."""
        torch = _torch()
        w = self._dev(white_noise, torch.float32).reshape(-1, self.n_cells)
        n = w.shape[0]
        if out is None:
            out = torch.empty((n, self.n_cells), dtype=torch.float32, device=self.device)
        st = self.stream if stream is None else C.c_void_p(stream.cuda_stream)
        _ffi.check(self._lib.ipp_generate_grf(self._h, n, self._ptr(w), self._ptr(out), st))
        self._keep_grf = w
        return out

    def generate_grf_rows(self, n: int, seed: int, subsequence: int, out, row_ids=None, row_offset: int = 0, stream=None,
                          group_rows: int = 0, group_subsequence=None) -> bool:
        """generate_grf with the white noise drawn inside the generator (the numbers normal_rows(.., n_cells, seed, subsequence, row_ids,
        row_offset) would have written).  group_rows / group_subsequence (host ints, <= 16 groups): fields of several episodes in one
        launch -- field i draws from subsequence + group_subsequence[i // group_rows]; a negative row id skips its field.  out=None:
        field i goes to the ALTERNATE ground-truth plane of env slot row_ids[i] (staged for its next episode; a folded reset with
        reset_gt=None flips the env to it).  Returns False when this grid has no such generator (caller: normal_rows + generate_grf)."""
        torch = _torch()
        ids = self._dev(row_ids, torch.int32)
        st = self.stream if stream is None else C.c_void_p(stream.cuda_stream)
        gs = None
        if group_rows:
            gs = (C.c_int64 * len(group_subsequence))(*[int(x) for x in group_subsequence])
        rc = self._lib.ipp_generate_grf_groups(self._h, int(n), int(group_rows), gs, self._ptr(ids), int(row_offset), int(seed) & (2 ** 64 - 1),
                                               int(subsequence) & (2 ** 64 - 1), self._ptr(out), st)
        if rc == -3:
            return False
        _ffi.check(rc)
        self._keep_grf = ids
        return True

    def step(self, actions, prev_actions, env_ids=None, dst_ids=None, meas_noise=None, *, cov_only=False,
             predict_only=False, adaptive=True, use_flight_time=True, given_observation=False, reward_out=None,
             status_out=None, update_prev=False, reset_src=None, reset_gt=None, init_action=None):
        """
        One batched env step.  reset_src / reset_gt / init_action (factor engines, in-place committed steps): the
        episode resets that fall on this step, folded into the launch (ipp_step_autoreset): reset_src [n] int32
        device tensor with the index into reset_gt ([k, H, W] float32 device tensor) or -1 per item.
        """
        torch = _torch()
        a = self._dev(actions, torch.float64).reshape(-1, 3)
        p = self._dev(prev_actions, torch.float64).reshape(-1, 3)
        n = a.shape[0]
        ids = self._dev(env_ids, torch.int32)
        dst = self._dev(dst_ids, torch.int32)
        nz = self._dev(meas_noise, torch.float32)
        if nz is not None:
            nz = nz.reshape(n, -1)
            if nz.shape[1] != self.meas_cap:
                pad = torch.zeros((n, self.meas_cap), dtype=torch.float32, device=self.device)
                pad[:, : min(nz.shape[1], self.meas_cap)] = nz[:, : self.meas_cap]
                nz = pad
        reward = reward_out if reward_out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        status = status_out if status_out is not None else torch.empty(n, dtype=torch.int32, device=self.device)
        flags = (_ffi.IPP_COV_ONLY if cov_only else 0) | (_ffi.IPP_PREDICT_ONLY if predict_only else 0) | \
                (_ffi.IPP_ADAPTIVE if adaptive else 0) | (_ffi.IPP_USE_FLIGHT_TIME if use_flight_time else 0) | \
                (_ffi.IPP_GIVEN_OBSERVATION if given_observation else 0) | (_ffi.IPP_UPDATE_PREV if update_prev else 0)
        if update_prev and (predict_only or not isinstance(prev_actions, torch.Tensor) or p.data_ptr() != prev_actions.data_ptr()):
            raise ValueError("update_prev needs prev_actions as a contiguous float64 device tensor and a committed step")
        if reset_src is not None:
            if dst is not None or predict_only:
                raise ValueError("reset_src: in-place committed steps only")
            if not isinstance(prev_actions, torch.Tensor) or p.data_ptr() != prev_actions.data_ptr():
                raise ValueError("reset_src needs prev_actions as a contiguous float64 device tensor (the reset writes it)")
            src = self._dev(reset_src, torch.int32)
            g = self._dev(reset_gt, torch.float32)  # (None: the fields were staged into the envs' alternate planes -- the reset flips)
            if src.numel() != n:
                raise ValueError("reset_src must have one entry per item")
            init = (C.c_double * 3)(*[float(x) for x in init_action])
            _ffi.check(self._lib.ipp_step_autoreset(self._h, self._ptr(ids), n, self._ptr(a), self._ptr(p), self._ptr(nz), flags,
                                                    self._ptr(reward), self._ptr(status), self._ptr(src), self._ptr(g), init,
                                                    self.stream))
            self._keep = (a, p, ids, nz, src, g)
            return reward, status
        _ffi.check(self._lib.ipp_step(self._h, self._ptr(ids), self._ptr(dst), n, self._ptr(a), self._ptr(p), self._ptr(nz),
                                      flags, self._ptr(reward), self._ptr(status), self.stream))
        self._keep = (a, p, ids, dst, nz)
        return reward, status

    def step_parts(self, actions, prev_actions, meas_noise, flags: int, reward, status, part_begin, streams, *,
                   reset_src=None, reset_gt=None, init_action=None):
        """ipp_step_parts: the whole batch as len(streams) launches, part p = positions [part_begin[p], part_begin[p + 1]) of
        the dispatch order (set_item_order) on streams[p] (torch streams).  All tensor arguments are preallocated device
        tensors (the batched driver's loop); nothing joins the streams here."""
        n = int(actions.shape[0])
        P = len(streams)
        key = (tuple(part_begin), tuple(st.cuda_stream for st in streams))
        cached = getattr(self, "_parts_c", None)
        if cached is None or cached[0] != key:  # (the batched driver passes the same partition and streams every step)
            cached = self._parts_c = (key, (C.c_int32 * (P + 1))(*[int(b) for b in part_begin]),
                                      (C.c_void_p * P)(*[C.c_void_p(st.cuda_stream) for st in streams]))
        begins, sts = cached[1], cached[2]
        init = (C.c_double * 3)(*[float(x) for x in init_action]) if init_action is not None else None
        _ffi.check(self._lib.ipp_step_parts(self._h, n, self._ptr(actions), self._ptr(prev_actions), self._ptr(meas_noise), int(flags),
                                            self._ptr(reward), self._ptr(status), self._ptr(reset_src), self._ptr(reset_gt), init,
                                            P, begins, sts))
        # Lifetime of the inputs: the launches run on the PART streams, the caching allocator only knows the caller's stream -- a tensor
        # the caller drops after this call could be handed out again while a part stream still reads it.  The tuples of the last
        # 2 x 8 calls are kept; a tuple is dropped only after an event recorded on every part stream BEHIND its launches has
        # completed (one event record per stream every 8 calls, one synchronize -- long complete -- before a half of the ring is
        # reused: ~1 us per call; record_stream on five tensors and two streams would be ~15 us of host time per step).
        ring = getattr(self, "_parts_ring", None)
        if ring is None or ring[0] != key[1]:
            torch = _torch()
            if ring is not None:
                # other streams from now on (rare): the old ring's tuples -- also those of the half that has not recorded its
                # events yet -- are read by launches on the OLD streams; join those before the tuples go
                for st in ring[5]:
                    st.synchronize()
            ring = self._parts_ring = [key[1], [None] * 16, [[torch.cuda.Event() for _ in streams] for _ in range(2)], 0, [False, False],
                                       list(streams)]
        i = ring[3]
        half = i // 8
        if i % 8 == 0 and ring[4][half]:
            for ev in ring[2][half]:
                ev.synchronize()  # the launches that read this half's tuples are done
        ring[1][i] = (actions, prev_actions, meas_noise, reset_src, reset_gt)
        if i % 8 == 7:
            for ev, st in zip(ring[2][half], streams):
                ev.record(st)
            ring[4][half] = True
        ring[3] = (i + 1) % 16

    def step_raw(self, n, actions, prev_actions, meas_noise, flags, reward, status, env_ids=None):
        """Zero-overhead variant for the benchmark loop: all arguments are preallocated device tensors."""
        _ffi.check(self._lib.ipp_step(self._h, self._ptr(env_ids), C.c_void_p(0), n, self._ptr(actions), self._ptr(prev_actions),
                                      self._ptr(meas_noise), flags, self._ptr(reward), self._ptr(status), self.stream))

    def observe(self, actions, env_ids=None, meas_noise=None):
        """Observation only (Sensor.take_measurement): returns (z [n, meas_cap], m [n], shape [n, 2]) device tensors."""
        torch = _torch()
        a = self._dev(actions, torch.float64).reshape(-1, 3)
        n = a.shape[0]
        ids = self._dev(env_ids, torch.int32)
        nz = self._dev(meas_noise, torch.float32)
        if nz is not None:
            nz = nz.reshape(n, -1)
            if nz.shape[1] != self.meas_cap:
                pad = torch.zeros((n, self.meas_cap), dtype=torch.float32, device=self.device)
                pad[:, : min(nz.shape[1], self.meas_cap)] = nz[:, : self.meas_cap]
                nz = pad
        z = torch.empty((n, self.meas_cap), dtype=torch.float32, device=self.device)
        m = torch.empty(n, dtype=torch.int32, device=self.device)
        shape = torch.empty((n, 2), dtype=torch.int32, device=self.device)
        _ffi.check(self._lib.ipp_observe(self._h, self._ptr(ids), n, self._ptr(a), self._ptr(nz), self._ptr(z), self._ptr(m),
                                         self._ptr(shape), self.stream))
        self._keep = (a, ids, nz)
        return z, m, shape

    def set_uav(self, max_v: float, max_a: float):
        _ffi.check(self._lib.ipp_set_uav(self._h, float(max_v), float(max_a)))

    def set_item_order(self, order):
        """Dispatch order of the items of the following step launches (device int32 permutation, kept alive by the caller's
        reference here; None = default).  Scheduling only: results do not depend on it."""
        if order is None:
            self._order = None
            _ffi.check(self._lib.ipp_set_item_order(self._h, None, 0))
            return
        o = self._dev(order, _torch().int32)
        self._order = o
        _ffi.check(self._lib.ipp_set_item_order(self._h, self._ptr(o), int(o.numel())))

    def set_reset_prior(self, prior):
        """Priors (sigma^2, l) of the episodes started by the following fused resets (step(reset_src=...)): device float64
        [k, 2] aligned with reset_gt, kept alive by the reference here; None = the config's prior (ipp_set_reset_prior)."""
        p = None if prior is None else self._dev(prior, _torch().float64)
        self._reset_prior = p
        _ffi.check(self._lib.ipp_set_reset_prior(self._h, self._ptr(p)))

    def set_adaptive(self, value_threshold: float, interval_factor: float):
        _ffi.check(self._lib.ipp_set_adaptive(self._h, float(value_threshold), float(interval_factor)))

    def fork(self, src_ids, dst_ids):
        torch = _torch()
        s, d = self._dev(src_ids, torch.int32), self._dev(dst_ids, torch.int32)
        _ffi.check(self._lib.ipp_fork(self._h, self._ptr(s), self._ptr(d), int(s.numel()), self.stream))
        self._keep = (s, d)

    def _read(self, fn, env, numel):
        torch = _torch()
        out = torch.empty(numel, dtype=torch.float32, device=self.device)
        _ffi.check(fn(self._h, int(env), self._ptr(out), self.stream))
        return out

    def read_mean(self, env):
        return self._read(self._lib.ipp_read_mean, env, self.n_cells).reshape(self.cfg.y_dim, self.cfg.x_dim)

    def read_diag(self, env):
        return self._read(self._lib.ipp_read_diag, env, self.n_cells)

    def read_gt(self, env):
        return self._read(self._lib.ipp_read_gt, env, self.n_cells).reshape(self.cfg.y_dim, self.cfg.x_dim)

    def read_cov(self, env):
        n = self.n_cells
        return self._read(self._lib.ipp_read_cov_dense, env, n * n).reshape(n, n)

    def rank(self, env) -> int:
        r = C.c_int32(0)
        _ffi.check(self._lib.ipp_read_rank(self._h, int(env), C.byref(r), self.stream))
        return int(r.value)

    def ranks(self, out=None):
        """Current factor rank of every slot as a device int32 tensor (no host sync)."""
        torch = _torch()
        if out is None:
            out = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
        _ffi.check(self._lib.ipp_read_ranks(self._h, self._ptr(out), self.stream))
        return out

    def write_mean(self, env, mean):
        t = self._dev(mean, _torch().float32, (self.n_cells,))
        _ffi.check(self._lib.ipp_write_mean(self._h, int(env), self._ptr(t), self.stream))
        self._keep = t

    def write_gt(self, env, gt):
        t = self._dev(gt, _torch().float32, (self.n_cells,))
        _ffi.check(self._lib.ipp_write_gt(self._h, int(env), self._ptr(t), self.stream))
        self._keep = t

    def write_cov(self, env, P):
        t = self._dev(P, _torch().float32, (self.n_cells, self.n_cells))
        _ffi.check(self._lib.ipp_write_cov_dense(self._h, int(env), self._ptr(t), self.stream))
        self._keep = t

    def metrics(self, env_ids=None, n: Optional[int] = None):
        torch = _torch()
        ids = self._dev(env_ids, torch.int32)
        n = int(ids.numel()) if ids is not None else int(n if n is not None else self.capacity)
        out = torch.empty((n, 8), dtype=torch.float32, device=self.device)
        _ffi.check(self._lib.ipp_metrics(self._h, self._ptr(ids), n, self._ptr(out), self.stream))
        self._keep = ids
        return out

    def normal(self, count: int, seed: int, subsequence: int = 0, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty(int(count), dtype=torch.float32, device=self.device)
        _ffi.check(self._lib.ipp_fill_normal(self._h, self._ptr(out), int(count), int(seed), int(subsequence), self.stream))
        return out

    def normal_rows(self, out, row_len: int, seed: int, subsequence: int, row_ids=None, row_offset: int = 0):
        """Row-keyed Philox normals (ipp_fill_normal_rows): out is a contiguous float32 device tensor [rows, row_len] or
        [planes, rows, row_len]; out[p, j, :] depends only on (seed, subsequence + p, row_ids[j] + row_offset)."""
        torch = _torch()
        if out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 device tensor")
        planes, rows = (1, out.shape[0]) if out.dim() == 2 else (out.shape[0], out.shape[1])
        if out.shape[-1] != row_len:
            raise ValueError("last dimension of out must be row_len")
        ids = self._dev(row_ids, torch.int32)
        if ids is not None and ids.numel() != rows:
            raise ValueError("row_ids needs one entry per row")
        _ffi.check(self._lib.ipp_fill_normal_rows(self._h, self._ptr(out), int(planes), int(rows), int(row_len), self._ptr(ids),
                                                  int(row_offset), int(seed) & (2 ** 64 - 1), int(subsequence) & (2 ** 64 - 1), self.stream))
        self._keep_rows = ids
        return out

    def probe_stream_pair(self, stream_a, stream_b, launches: int = 12) -> float:
        """ms that `launches` dependent 30-us kernels on each of the two (torch) streams take when issued alternately
        (ipp_probe_stream_pair): small = the streams' hardware queues dispatch independently."""
        ms = C.c_double(0.0)
        _ffi.check(self._lib.ipp_probe_stream_pair(self._h, C.c_void_p(stream_a.cuda_stream), C.c_void_p(stream_b.cuda_stream),
                                                   int(launches), C.byref(ms)))
        return float(ms.value)

    def debug_capture(self, enable: bool = True):
        """The steps that follow keep fp64 copies of S, L^-1, z and y per item for debug_item (off by default: 1.4 KB of stores per item)."""
        _ffi.check(self._lib.ipp_debug_capture(self._h, 1 if enable else 0))

    def debug_item(self, idx: int) -> Dict:
        it = _ffi.IppStepItem()
        _ffi.check(self._lib.ipp_debug_step_item(self._h, int(idx), C.byref(it), self.stream))
        m = it.m
        return dict(
            env=it.env, dst=it.dst, rank_before=it.rank_before, status=it.status,
            fov=(it.xl, it.xr, it.yu, it.yd), rf=it.rf, m=m, f=it.f, cost=it.cost, noise_var=it.noise_var,
            S=np.array(it.S[: m * m]).reshape(m, m), Linv=np.array(it.Linv[: m * m]).reshape(m, m),
            z=np.array(it.z[:m]), y=np.array(it.y[:m]),
        )

    def streamed_bytes(self, reset: bool = True) -> int:
        """Bytes the gain kernel streamed / wrote since the last reset (device counter; synchronises)."""
        b = C.c_uint64(0)
        _ffi.check(self._lib.ipp_streamed_bytes(self._h, C.byref(b), 1 if reset else 0, self.stream))
        return int(b.value)

    def streamed_bytes_detail(self, reset: bool = True):
        """(algorithmic bytes per SURVEY 8(d): r + m + 4 floats per touched cell, bytes of the fused kernel's second read of
        mean / diag for the mask) since the last reset; synchronises."""
        b, x = C.c_uint64(0), C.c_uint64(0)
        _ffi.check(self._lib.ipp_streamed_bytes_detail(self._h, C.byref(b), C.byref(x), 1 if reset else 0, self.stream))
        return int(b.value), int(x.value)

    def streamed_bytes_needed(self) -> int:
        """Of the last streamed_bytes(_detail) read: the bytes on the lanes inside the stored columns' own rectangles (patch engines)."""
        b = C.c_uint64(0)
        fn = getattr(self._lib, "ipp_streamed_bytes_needed", None) if os.environ.get("IPP_AB_OLD_LIB") else self._lib.ipp_streamed_bytes_needed
        if fn is None or fn.argtypes is None:  # (tools/ab_libs.sh timing an older build of the engine)
            return 0
        _ffi.check(fn(self._h, C.byref(b)))
        return int(b.value)

    def profile(self, enable: bool):
        _ffi.check(self._lib.ipp_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, kind: int, reset: bool = True):
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        _ffi.check(self._lib.ipp_profile_read(self._h, int(kind), C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return float(ms.value), int(cnt.value)

    def profile_read_busy(self, kind: int, reset: bool = True):
        """(ms during which at least one kernel of that kind ran, launches) since the last reset of this figure."""
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        _ffi.check(self._lib.ipp_profile_read_busy(self._h, int(kind), C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return float(ms.value), int(cnt.value)

    # ------------------------------------------------------------------ byte accounting
    def algorithmic_bytes_per_step(self, rank_before: float, m: float) -> float:
        """fp32 bytes one committed step must move: factor 4N(r+m)+16N, dense 8N^2+16N."""
        n = self.n_cells
        if self.state == "factor":
            return 4.0 * n * (rank_before + m) + 16.0 * n
        return 8.0 * n * n + 16.0 * n
