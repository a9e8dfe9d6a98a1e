"""
ctypes binding of the C-ABI in include/ipp_engine.h (lib/libipp_hip.so, built by csrc/Makefile).

There is no CPU fallback: if the HIP library is missing or does not load, importing the engine fails
loudly (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C ipp-rl_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPP_HIP_LIB") or os.path.join(_HERE, "lib", "libipp_hip.so")  # override: A/B builds only

IPP_DENSE, IPP_FACTOR = 0, 1
IPP_COV_ONLY, IPP_PREDICT_ONLY, IPP_ADAPTIVE, IPP_USE_FLIGHT_TIME, IPP_GIVEN_OBSERVATION, IPP_UPDATE_PREV = 1, 2, 4, 8, 16, 32
STATUS_OK, STATUS_CHOL_FALLBACK, STATUS_NOT_PD, STATUS_RANK_FULL, STATUS_BAD_FOOTPRINT = 0, 1, 2, 3, 4
IPP_MAX_MEAS = 25
ABI_VERSION = 14
AB_MIN_ABI = 13  # oldest library tools/ab_kernels.py may load under IPP_AB_OLD_LIB (v14 added the ipp_arena_* calls, nothing else)
IPP_ARENA_HIPMALLOC, IPP_ARENA_VMM = 0, 1


class IppConfig(C.Structure):
    _fields_ = [
        ("x_dim", C.c_int32), ("y_dim", C.c_int32),
        ("resolution", C.c_double),
        ("tan_half_fov_x", C.c_double), ("tan_half_fov_y", C.c_double),
        ("rf_altitude", C.c_double),
        ("coeff_a", C.c_double), ("coeff_b", C.c_double),
        ("signal_variance", C.c_double), ("length_scale", C.c_double),
        ("max_v", C.c_double), ("max_a", C.c_double),
        ("value_threshold", C.c_double), ("interval_factor", C.c_double),
        ("cluster_radius", C.c_double),
        ("state_repr", C.c_int32), ("capacity", C.c_int32), ("rank_cap", C.c_int32), ("max_batch", C.c_int32),
        ("max_measurements", C.c_int32), ("tile_threads", C.c_int32), ("window_rows", C.c_int32), ("score_scratch", C.c_int32), ("node_capacity", C.c_int32), ("fixed_prior", C.c_int32),
    ]


class IppInfo(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_cells", C.c_int32), ("n_pad", C.c_int32), ("tile_threads", C.c_int32),
        ("n_tiles", C.c_int32), ("meas_cap", C.c_int32), ("fp_cap", C.c_int32), ("window_rows", C.c_int32),
        ("arena_bytes", C.c_uint64), ("cov_slot_bytes", C.c_uint64), ("step_lds_bytes", C.c_uint64),
        ("fused_step", C.c_int32), ("patch_layout", C.c_int32), ("patch_waves", C.c_int32), ("patch_big_min_items", C.c_int32),
        ("patch_split_min_items", C.c_int32), ("patch_two_wave_min_items", C.c_int32),
    ]


class IppStepItem(C.Structure):
    _fields_ = [
        ("env", C.c_int32), ("dst", C.c_int32), ("rank_before", C.c_int32), ("status", C.c_int32),
        ("xl", C.c_int32), ("xr", C.c_int32), ("yu", C.c_int32), ("yd", C.c_int32),
        ("rf", C.c_int32), ("m", C.c_int32), ("f", C.c_int32), ("pad", C.c_int32),
        ("cost", C.c_double), ("noise_var", C.c_double),
        ("S", C.c_double * (IPP_MAX_MEAS * IPP_MAX_MEAS)),
        ("Linv", C.c_double * (IPP_MAX_MEAS * IPP_MAX_MEAS)),
        ("z", C.c_double * IPP_MAX_MEAS),
        ("y", C.c_double * IPP_MAX_MEAS),
    ]


_P = C.c_void_p


class IppMctsTables(C.Structure):
    """ipp_mcts_tables (include/ipp_engine.h): dimensions, search constants and the [dev] buffers of a device-side tree search."""
    _fields_ = [
        ("roots", C.c_int32), ("kmax", C.c_int32), ("nodes_per_root", C.c_int32), ("dev_per_root", C.c_int32),
        ("table_size", C.c_int32), ("max_depth", C.c_int32), ("wave", C.c_int32), ("horizon", C.c_int32),
        ("grid_w", C.c_int32), ("grid_h", C.c_int32), ("n_levels", C.c_int32), ("n_off", C.c_int32),
        ("num_actions", C.c_int32), ("use_flight_time", C.c_int32), ("tie_break", C.c_int32), ("device", C.c_int32),
        ("res", C.c_double), ("max_dist", C.c_double), ("gamma", C.c_double), ("puct_init", C.c_double),
        ("puct_base", C.c_double), ("fpf", C.c_double), ("vmax", C.c_double), ("amax", C.c_double),
        ("actions", _P), ("cell_action", _P), ("off_x", _P), ("off_y", _P),
        ("zkey", _P), ("uniform_ps", _P), ("t_idx", _P), ("t_ps", _P),
        ("t_nsa", _P), ("t_qsa", _P), ("t_num", _P), ("t_child", _P),
        ("n_k", _P), ("n_ns", _P), ("n_flags", _P), ("n_hash", _P),
        ("n_value", _P), ("n_devpath", _P), ("root_count", _P), ("dev_count", _P),
        ("h_keys", _P), ("h_vals", _P), ("p_node", _P), ("p_k", _P),
        ("p_cost", _P), ("p_len", _P), ("leaf", _P), ("pend_node", _P),
        ("pend_depth", _P), ("pend_sim", _P), ("pend_prev", _P), ("pend_budget", _P),
        ("pend_count", _P), ("rq_root", _P), ("rq_parent", _P), ("rq_k", _P),
        ("rq_child", _P), ("rq_newdev", _P), ("rq_cost", _P), ("rq_prev", _P),
        ("rq_action", _P), ("rq_count", _P), ("ts_paths", _P), ("ts_reward", _P),
        ("ts_status", _P), ("err", _P),
        ("puct_c", _P), ("sqrt_ns1", _P), ("ns_table_n", C.c_int64),
        ("root_base", C.c_int32), ("dev_base", C.c_int32), ("scratch_base", C.c_int32), ("reserved0", C.c_int32),
    ]

# name -> (restype, argtypes); exactly the symbols include/ipp_engine.h declares
PROTOTYPES = {
    "ipp_abi_version": (C.c_int, []),
    "ipp_min_window_rows": (C.c_int, [_P, _P]),
    "ipp_last_error": (C.c_char_p, []),
    "ipp_engine_arena_bytes": (C.c_int, [C.POINTER(IppConfig), C.POINTER(C.c_uint64)]),
    "ipp_engine_create": (C.c_int, [C.POINTER(IppConfig), C.c_int, _P, C.c_uint64, C.POINTER(_P)]),
    "ipp_engine_destroy": (C.c_int, [_P]),
    "ipp_engine_info": (C.c_int, [_P, C.POINTER(IppInfo)]),
    "ipp_reset": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P]),
    "ipp_reset_episode": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "ipp_score_actions": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_uint32, _P, _P, _P]),
    "ipp_state_plane": (C.c_int, [_P, C.c_int32, _P, C.c_uint32, _P, _P]),
    "ipp_tree_step": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P, C.c_uint32, _P, _P, _P]),
    "ipp_tree_read_diag": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_mcts_select": (C.c_int, [C.POINTER(IppMctsTables), _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _P]),
    "ipp_mcts_steps": (C.c_int, [_P, C.POINTER(IppMctsTables), C.c_int32, C.c_int32, C.c_uint32, _P]),
    "ipp_mcts_expand": (C.c_int, [C.POINTER(IppMctsTables), _P, _P, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_uint64, _P]),
    "ipp_mcts_backup": (C.c_int, [C.POINTER(IppMctsTables), C.c_int32, _P]),
    "ipp_mcts_policy": (C.c_int, [C.POINTER(IppMctsTables), _P, C.c_double, C.c_int32, _P, _P, _P, _P]),
    "ipp_tree_score_actions": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, C.c_uint32, _P, _P, _P]),
    "ipp_generate_grf": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "ipp_generate_grf_rows": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_uint64, C.c_uint64, _P, _P]),
    "ipp_generate_grf_groups": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int64, C.c_uint64, C.c_uint64, _P, _P]),
    "ipp_step": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, C.c_uint32, _P, _P, _P]),
    "ipp_step_autoreset": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "ipp_step_parts": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    "ipp_observe": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "ipp_set_uav": (C.c_int, [_P, C.c_double, C.c_double]),
    "ipp_set_adaptive": (C.c_int, [_P, C.c_double, C.c_double]),
    "ipp_set_item_order": (C.c_int, [_P, _P, C.c_int32]),
    "ipp_set_reset_prior": (C.c_int, [_P, _P]),
    "ipp_fork": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "ipp_read_mean": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_read_diag": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_read_gt": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_read_cov_dense": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_read_rank": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), _P]),
    "ipp_read_ranks": (C.c_int, [_P, _P, _P]),
    "ipp_write_mean": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_write_gt": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_write_cov_dense": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ipp_metrics": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "ipp_fill_normal": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint64, _P]),
    "ipp_fill_normal_rows": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "ipp_probe_stream_pair": (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(C.c_double)]),
    "ipp_arena_alloc": (C.c_int, [C.c_int, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(_P)]),
    "ipp_arena_free": (C.c_int, [_P]),
    "ipp_arena_retired_bytes": (C.c_int, [C.POINTER(C.c_uint64)]),
    "ipp_arena_trim": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "ipp_arena_latency": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    "ipp_arena_probe": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    "ipp_debug_capture": (C.c_int, [_P, C.c_int32]),
    "ipp_debug_step_item": (C.c_int, [_P, C.c_int32, C.POINTER(IppStepItem), _P]),
    "ipp_streamed_bytes": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int32, _P]),
    "ipp_streamed_bytes_detail": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32, _P]),
    "ipp_streamed_bytes_needed": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ipp_profile_enable": (C.c_int, [_P, C.c_int32]),
    "ipp_profile_read": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "ipp_profile_read_busy": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
}

_lib = None


class IppError(RuntimeError):
    pass


def load():
    """Load libipp_hip.so once; raise (never fall back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IppError(
            f"HIP engine library not found at {LIB_PATH}: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # The engine shares the HIP runtime of the process with PyTorch (device memory, streams).  PyTorch's wheel carries
    # its own libamdhip64: it has to be the first HIP runtime the process loads, otherwise the engine binds to
    # /opt/rocm's copy and the process ends up with two runtimes, one of which sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:  # symbol checks on a machine without torch
        pass
    lib = C.CDLL(LIB_PATH)
    # tools/ab_kernels.py may time an OLDER build of the library against this binding (IPP_AB_OLD_LIB=1): only a library whose
    # ABI is at least AB_MIN_ABI (struct layouts and call signatures unchanged since then; newer entry points may be absent
    # and are then not bound) and only when the library is not the in-tree product build -- never a general switch
    lib.ipp_abi_version.restype, lib.ipp_abi_version.argtypes = C.c_int, []
    abi = lib.ipp_abi_version()
    ab_old = bool(os.environ.get("IPP_AB_OLD_LIB")) and bool(os.environ.get("IPP_HIP_LIB")) and AB_MIN_ABI <= abi <= ABI_VERSION
    if abi != ABI_VERSION and not ab_old:
        raise IppError(f"libipp_hip.so ABI {abi} != binding ABI {ABI_VERSION}: rebuild")
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        except AttributeError:
            if ab_old:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise IppError(f"ipp engine error {rc}: {load().ipp_last_error().decode(errors='replace')}")
