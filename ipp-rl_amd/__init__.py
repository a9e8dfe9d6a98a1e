"""
ipp_rl_amd -- MI355X-native batched IPP environment-step engine behind the Mapping / Sensor /
Simulation class surface of dmar-bonn/ipp-rl.

    ipp_rl_amd.engine      IPPEngine: ctypes handle on the HIP C-ABI (include/ipp_engine.h)
    ipp_rl_amd.vec_env     VecIPPEnv: batched env driver (B >> 1, opaque env slots)
    ipp_rl_amd.mapping / sensors / simulations / planning.common
                           drop-in counterparts of the reference modules of the same name

The compute path is HIP only (csrc/, built into lib/libipp_hip.so); nothing here falls back to a CPU
implementation.
"""
from ._ffi import IppError  # noqa: F401
from .engine import EngineConfig, IPPEngine  # noqa: F401

__all__ = ["EngineConfig", "IPPEngine", "IppError"]
