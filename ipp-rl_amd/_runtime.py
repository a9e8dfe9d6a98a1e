"""
Single-environment compatibility runtime: the drop-in Mapping / Sensor / Simulation classes keep the
reference's NumPy-in / NumPy-out surface (fp64 arrays the planners index, hash and assign), and route every
map operation through a small dense-state HIP engine (2 slots) looked up here.  Nothing in this module
computes on the CPU; without the HIP library or a GPU the first call raises.

Engines are cached per configuration and are never part of an object's pickled state (the reference pickles
Mapping objects into multiprocessing workers and experiment.pkl: SURVEY 8(b)).

Processes.  A SPAWNED worker (multiprocessing.get_context("spawn"), torch.multiprocessing.spawn) is a fresh process:
it unpickles the Mapping and opens its own engine lazily on first use.  A FORKED worker of a parent that has already
opened an engine cannot use the GPU at all: HIP state does not survive fork() and PyTorch refuses to re-initialise
the device in a forked child.  The child therefore abandons the inherited engines WITHOUT destroying them (their
streams / events / device memory belong to the parent's runtime; calling hipStreamDestroy on them from the child can
hang) and the first map operation raises an IppError that says to use the spawn start method -- the reference's
`Pool(4)` in greedy_search (planning/common/optimization.py:86-90) and its self-play pools fork by default on Linux,
so callers set `multiprocessing.set_start_method("spawn")` once.  A fork BEFORE any engine exists
is harmless: the child initialises the GPU itself.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

from . import _ffi
from . import engine as _engine_mod
from ._device_array import DeviceCov, SlotStore
from .engine import EngineConfig, IPPEngine

_ENGINES: Dict[Tuple, IPPEngine] = {}
_STORES: Dict[int, "SlotStore"] = {}  # id(engine) -> device state slots handed out as DeviceCov objects
STATE_SLOTS = int(os.environ.get("IPP_COMPAT_SLOTS", "16"))  # most dense states kept on the device per compat engine
STATE_BYTES = int(float(os.environ.get("IPP_COMPAT_STATE_GB", "4")) * (1 << 30))  # ... and the device memory they may take


def state_slots_for(n_cells: int) -> int:
    """Dense state slots of a compat engine: IPP_COMPAT_SLOTS (16) while they fit IPP_COMPAT_STATE_GB (4 GB), at least 2: a
    slot is N x Npad fp32 -- 25 MB at 50x50, 0.4 GB at 100x100, 6.4 GB at 200x200 (16 of those would be 115 GB)."""
    per_slot = 4 * n_cells * (n_cells + 1024)
    return int(max(2, min(STATE_SLOTS, STATE_BYTES // max(1, per_slot))))


def config_key(cfg: EngineConfig) -> Tuple:
    return (cfg.x_dim, cfg.y_dim, cfg.resolution, cfg.angle_x, cfg.angle_y, cfg.rf_altitude, cfg.coeff_a, cfg.coeff_b,
            cfg.signal_variance, cfg.length_scale, cfg.cluster_radius)


def compat_engine(cfg: EngineConfig) -> IPPEngine:
    """Dense-state engine for `cfg`: slot 0 working map, slot 1 scratch (ground truth / observations), slots 2.. hold the
    map states the planners keep alive (DeviceCov, _device_array.py)."""
    if _engine_mod.forked_with_gpu():  # (engine.py abandons every inherited engine at fork, without destroying it)
        raise _ffi.IppError(_engine_mod.FORK_MESSAGE)
    key = config_key(cfg)
    eng = _ENGINES.get(key)
    if eng is None:
        m_cap = 9 if cfg.resolution >= 2 else 25
        slots = state_slots_for(cfg.n_cells)
        try:
            eng = IPPEngine(cfg, capacity=2 + slots, state="dense", rank_cap=1, max_batch=2, max_measurements=m_cap,
                            score_scratch=True)
        except Exception as exc:  # (the arena is one torch allocation: an out-of-memory error lands here)
            raise _ffi.IppError(f"compat engine for a {cfg.x_dim}x{cfg.y_dim} map ({2 + slots} dense state slots of "
                                f"{4 * cfg.n_cells * cfg.n_cells / 1e9:.2f} GB) could not be created: {exc}; lower "
                                f"IPP_COMPAT_SLOTS (now {STATE_SLOTS}) or IPP_COMPAT_STATE_GB (now {STATE_BYTES / (1 << 30):g})") from exc
        _ENGINES[key] = eng
        _STORES[id(eng)] = SlotStore(eng, 2, slots)
    return eng


def state_store(eng: IPPEngine) -> "SlotStore":
    return _STORES[id(eng)]


def on_device(eng: IPPEngine, state) -> DeviceCov:
    """`state` as a DeviceCov of `eng` with a live slot: DeviceCov objects of this engine are used where they are, anything
    else (plain float64 matrices, states of another engine) is uploaded into a fresh slot."""
    store = state_store(eng)
    if isinstance(state, DeviceCov) and state.device_slot(store) is not None:
        return state
    host = np.ascontiguousarray(np.asarray(state), dtype=np.float64)
    obj = DeviceCov(store, host.shape[0], host=host)
    obj.device_slot(store)
    return obj


def engine_config_from(grid_map, sensor, signal_variance: float, length_scale: float, cluster_radius=None) -> EngineConfig:
    """cluster_radius: the simulation passes its own (it builds its first ground truth BEFORE it is attached to the
    sensor: planning/ipp_mission_node.py:40-42, simulations/simulations.py:41); the mapping takes the attached one."""
    cluster = cluster_radius
    if cluster is None:
        cluster = getattr(getattr(sensor, "sensor_simulation", None), "cluster_radius", None)
    return EngineConfig(
        x_dim=int(grid_map.x_dim), y_dim=int(grid_map.y_dim), resolution=float(grid_map.resolution),
        angle_x=float(sensor.angle_x), angle_y=float(sensor.angle_y),
        coeff_a=float(sensor.sensor_model.coeff_a), coeff_b=float(sensor.sensor_model.coeff_b),
        signal_variance=float(signal_variance), length_scale=float(length_scale),
        cluster_radius=float(cluster) if cluster is not None else 5.0,
    )


def to_host64(t) -> np.ndarray:
    """fp64 host copy of a device tensor; large tensors are widened on the device (cheaper than numpy's astype)."""
    t = t.detach()
    if t.numel() >= (1 << 16):
        return t.double().cpu().numpy()
    return t.cpu().numpy().astype(np.float64)
