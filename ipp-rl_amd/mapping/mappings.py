"""
Mapping: drop-in for the reference's mapping/mappings.py (Mapping.update_grid_map, init_priors,
kalman_filter_update) with the arithmetic on the GPU.

  * update_grid_map -> one fused HIP step on the dense compat engine (include/ipp_engine.h: ipp_step with
    IPP_GIVEN_OBSERVATION / IPP_COV_ONLY, out of place into a new state slot), fp32 state on the device; covariances
    are returned as DeviceCov (_device_array.py): array-likes that stay on the GPU until a caller looks at the whole
    matrix (results agree with the reference within 1e-5, tests/test_hip_classes.py).
  * init_priors GP branch -> ipp_reset (analytic Matern-3/2 == the unfitted GPR the reference builds,
    mapping/mappings.py:242-258); the shuffle_prior_cov draws stay on NumPy's legacy stream (:239-240).
  * the non-GP prior (:219-233) and the generic dense-H kalman_filter_update (:156-215, no caller in the
    reference) are plain library GEMMs: torch (rocBLAS) on the device.
"""
import logging
from typing import Optional, Tuple

import numpy as np

from .. import _runtime
from .._device_array import DeviceCov
from .grid_maps import GridMap, _require

logger = logging.getLogger(__name__)


class Mapping:
    def __init__(self, grid_map: GridMap, sensor, shuffle_prior_cov: bool = False):
        self.grid_map = grid_map
        self.sensor = sensor
        self.shuffle_prior_cov = shuffle_prior_cov
        self._prior_scale = None  # (sigma^2, l) actually used by init_priors
        self.init_priors()

    # ---- validated config getters (reference mapping/mappings.py:23-112)
    def _mapping_param(self, key: str):
        return _require(self.grid_map.params, "mapping", key, f"mapping's '{key}'")

    signal_variance = property(lambda self: self._mapping_param("signal_variance"))
    noise_variance = property(lambda self: self._mapping_param("noise_variance"))
    length_scale = property(lambda self: self._mapping_param("length_scale"))
    nu = property(lambda self: self._mapping_param("nu"))
    fit_gaussian_process = property(lambda self: self._mapping_param("fit_gaussian_process"))
    prior_cov_mean = property(lambda self: self._mapping_param("prior_cov_mean"))
    prior_cov_std = property(lambda self: self._mapping_param("prior_cov_std"))

    # ---- engine plumbing (never pickled: looked up per call)
    def _engine(self):
        sv, ls = self._prior_scale if self._prior_scale is not None else (self.signal_variance, self.length_scale)
        cfg = _runtime.engine_config_from(self.grid_map, self.sensor, self.signal_variance, self.length_scale)
        eng = _runtime.compat_engine(cfg)
        return eng, (sv, ls)

    def init_priors(self):
        """Prior mean 0.5 and prior covariance (reference mapping/mappings.py:217-261)."""
        gm = self.grid_map
        n = gm.num_grid_cells
        if not self.fit_gaussian_process:
            import torch

            mu, sd = self.prior_cov_mean, self.prior_cov_std
            if self.shuffle_prior_cov:
                mu = np.random.uniform(low=0.1, high=self.prior_cov_mean)
                sd = mu
            gm.mean = 0.5 * np.ones((gm.y_dim, gm.x_dim))
            A = np.random.normal(mu, sd, (n, n))  # host legacy stream: seed parity (:226-228)
            eng, _ = self._engine()
            At = torch.as_tensor(A, dtype=torch.float64, device=eng.device)
            gm.cov_matrix = ((At @ At.T) / torch.linalg.norm(At, ord="fro")).cpu().numpy()
            return
        if float(self.nu) != 1.5:
            logger.error("Only the Matern nu=1.5 prior of config/example.yaml is implemented on the device")
            raise ValueError
        sv, ls = self.signal_variance, self.length_scale
        if self.shuffle_prior_cov:
            sv = np.random.uniform(low=0.8 * self.signal_variance, high=1.2 * self.signal_variance)
            ls = np.random.uniform(low=0.8 * self.length_scale, high=1.2 * self.length_scale)
        self._prior_scale = (float(sv), float(ls))
        eng, _ = self._engine()
        prior = DeviceCov.new_on_device(_runtime.state_store(eng), n)  # the prior is built in its slot and stays there
        eng.reset(env_ids=[prior.device_slot(_runtime.state_store(eng))], prior_scale=np.array([[sv, ls]]))
        gm.mean = 0.5 * np.ones((gm.y_dim, gm.x_dim))
        gm.cov_matrix = prior

    def update_grid_map(
        self,
        measurement_position: np.array,
        measurement_data: np.array = None,
        cov_only: bool = False,
        predict_only: bool = False,
        current_cov_matrix: np.array = None,
    ):
        """
        Kalman update of the map from a measurement at `measurement_position` (reference
        mapping/mappings.py:114-153).  predict_only -> returns (x or None, P') and leaves the map untouched;
        otherwise assigns grid_map.mean / grid_map.cov_matrix.
        """
        gm = self.grid_map
        P_in = gm.cov_matrix if current_cov_matrix is None else current_cov_matrix
        want_mean = not cov_only
        if want_mean and measurement_data is None:
            raise AttributeError("measurement_data is required unless cov_only=True")  # reference: None.flatten()
        eng, _ = self._engine()
        store = _runtime.state_store(eng)
        src = _runtime.on_device(eng, P_in)  # states this layer handed out are still on the device: no upload
        src._pinned = True
        try:
            P_new = DeviceCov.new_on_device(store, src.shape[0])  # P' is a new array in the reference too (:190)
            s_slot, d_slot = src.device_slot(store), P_new.device_slot(store)
            if want_mean:
                eng.write_mean(s_slot, gm.mean)
            pos = np.asarray(measurement_position, dtype=np.float64).reshape(1, 3)
            z = None if not want_mean else np.asarray(measurement_data, dtype=np.float64).reshape(1, -1)
            _, status = eng.step(pos, pos, env_ids=[s_slot], dst_ids=[d_slot], meas_noise=z, cov_only=not want_mean,
                                 adaptive=False, use_flight_time=False, given_observation=want_mean)
            st = int(status[0])
        finally:
            src._pinned = False
        if st == 1:
            logger.error("Cholesky decomposition failed: S is not positive definite")
            logger.info("Fallback to classical matrix inversion")
        elif st != 0:
            logger.error(f"HIP step rejected the measurement footprint (status {st})")
            raise ValueError
        x_new = _runtime.to_host64(eng.read_mean(d_slot)) if want_mean else None
        if predict_only:
            return (None, P_new) if cov_only else (x_new.reshape(gm.mean.shape), P_new)
        gm.mean = x_new.reshape(gm.mean.shape)  # reference fails here too when cov_only and not predict_only
        gm.cov_matrix = P_new

    @staticmethod
    def kalman_filter_update(
        P: np.array,
        H: np.array,
        R: np.array,
        grid_mean: np.array = None,
        observation: np.array = None,
        cov_only: bool = False,
    ) -> Tuple[Optional[np.array], np.array]:
        """
        Generic dense-H Kalman update kept for API compatibility (reference mapping/mappings.py:156-215; it has
        no caller outside update_grid_map there).  Library linear algebra on the device, fp64.
        """
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("kalman_filter_update needs a HIP device; there is no CPU fallback")
        dev = torch.device("cuda")
        Pt = torch.as_tensor(np.asarray(P), dtype=torch.float64, device=dev)
        Ht = torch.as_tensor(np.asarray(H), dtype=torch.float64, device=dev)
        Rt = torch.as_tensor(np.asarray(R), dtype=torch.float64, device=dev)
        PHt = Pt @ Ht.T
        S = Ht @ PHt + Rt
        S = 0.5 * (S + S.T)
        L, info = torch.linalg.cholesky_ex(S)
        if int(info) == 0:
            L_inv = torch.linalg.inv(L.T)
            Wc = PHt @ L_inv
            P_new = Pt - Wc @ Wc.T
            gain = Wc @ L_inv.T
        else:
            logger.error("Cholesky decomposition failed: S is not positive definite")
            logger.info("Fallback to classical matrix inversion")
            S_inv = torch.linalg.inv(S)
            P_new = Pt - PHt @ (S_inv @ PHt.T)
            gain = PHt @ S_inv
        if cov_only:
            return None, P_new.cpu().numpy()
        x = torch.as_tensor(np.asarray(grid_mean).flatten(order="C"), dtype=torch.float64, device=dev)
        zt = torch.as_tensor(np.asarray(observation).flatten(order="C"), dtype=torch.float64, device=dev)
        x = x + gain @ (zt - Ht @ x)
        return x.cpu().numpy(), P_new.cpu().numpy()
