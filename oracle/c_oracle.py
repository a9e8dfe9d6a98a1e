"""
ctypes wrapper of oracle/libipp_oracle.so (plain-C fp64 restatement) -- TEST INFRASTRUCTURE ONLY.
Used by tests/ and by bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("IPP_ORACLE_LIB") or os.path.join(_HERE, "libipp_oracle.so")  # override: `make -C oracle asan`
OC_MAX_M = 25
COV_ONLY, PREDICT_ONLY, ADAPTIVE, USE_FLIGHT_TIME = 1, 2, 4, 8


class OcConfig(C.Structure):
    _fields_ = [("x_dim", C.c_int), ("y_dim", C.c_int), ("resolution", C.c_double), ("tan_half_fov_x", C.c_double),
                ("tan_half_fov_y", C.c_double), ("rf_altitude", C.c_double), ("coeff_a", C.c_double),
                ("coeff_b", C.c_double), ("max_v", C.c_double), ("max_a", C.c_double),
                ("value_threshold", C.c_double), ("interval_factor", C.c_double)]


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oc_step.restype = C.c_int
        _lib.oc_run_batch.restype = C.c_int
        _lib.oc_max_threads.restype = C.c_int
    return _lib


def make_config(cfg) -> OcConfig:
    """cfg: oracle.ipp_oracle.OracleConfig (or anything with the same attribute names)."""
    return OcConfig(cfg.x_dim, cfg.y_dim, cfg.resolution, float(np.tan(0.5 * np.radians(cfg.angle_x))),
                    float(np.tan(0.5 * np.radians(cfg.angle_y))), 10.0, cfg.coeff_a, cfg.coeff_b, cfg.max_v, cfg.max_a,
                    cfg.value_threshold, cfg.interval_factor)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def matern_prior(cfg, sv=None, ls=None):
    n = cfg.x_dim * cfg.y_dim
    P = np.empty((n, n))
    c = make_config(cfg)
    lib().oc_matern_prior(C.byref(c), C.c_double(cfg.signal_variance if sv is None else sv),
                          C.c_double(cfg.length_scale if ls is None else ls), _p(P))
    return P


def step(cfg, P, mean, gt, action, prev, eps=None, flags=ADAPTIVE | USE_FLIGHT_TIME):
    """In-place fused step on (P, mean); returns (rc, reward, z)."""
    n = cfg.x_dim * cfg.y_dim
    c = make_config(cfg)
    reward = C.c_double(0)
    z = np.zeros(OC_MAX_M)
    scratch = np.empty(2 * n * OC_MAX_M)
    a = np.ascontiguousarray(action, dtype=np.float64)
    p = np.ascontiguousarray(prev, dtype=np.float64)
    e = None if eps is None else np.ascontiguousarray(np.concatenate([np.ravel(eps), np.zeros(OC_MAX_M)])[:OC_MAX_M])
    rc = lib().oc_step(C.byref(c), _p(P), _p(mean), _p(gt), _p(a), _p(p), _p(e), C.c_int(flags), C.byref(reward), _p(z),
                       _p(scratch))
    return rc, reward.value, z


def grf_from_kernel(white, h):
    H, W = white.shape
    out = np.empty((H, W))
    lib().oc_grf_from_kernel(C.c_int(H), C.c_int(W), _p(np.ascontiguousarray(white)), _p(np.ascontiguousarray(h)), _p(out))
    return out


def run_batch(cfg, P_all, mean_all, gt_all, actions, init_prev, eps=None, flags=ADAPTIVE | USE_FLIGHT_TIME, threads=0):
    """actions [steps][B][3]; eps [steps][B][OC_MAX_M] or None; returns rewards [steps][B]."""
    steps, B = actions.shape[0], actions.shape[1]
    c = make_config(cfg)
    rewards = np.zeros((steps, B))
    rc = lib().oc_run_batch(C.byref(c), C.c_int(B), C.c_int(steps), _p(P_all), _p(mean_all), _p(gt_all),
                            _p(np.ascontiguousarray(actions, dtype=np.float64)),
                            _p(np.ascontiguousarray(init_prev, dtype=np.float64)), _p(eps), C.c_int(flags), _p(rewards),
                            C.c_int(threads))
    if rc < 0:
        raise RuntimeError(f"oc_run_batch failed: {rc}")
    return rewards


def max_threads() -> int:
    return int(lib().oc_max_threads())
