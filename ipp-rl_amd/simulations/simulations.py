"""
ScalarFieldSimulation / GaussianRandomField (reference simulations/simulations.py:17-47) on the device.

take_measurement: the standard normals come from NumPy's legacy global stream (one normal(size=shape) draw per
measurement, like sensor_manipulations.py:56-57, so seeded runs consume the stream identically); crop,
INTER_AREA downsample, scaling by the noise "variance", add and clip run in the HIP prologue kernel
(ipp_observe).  create_ground_truth_map: white noise from the same stream, FFT-filtered field on the device
(ipp_reset).  Hotspot / Split / Temperature generators of the reference are host-side dataset generators
outside the hot path and are not provided.
"""
import logging
import math

import numpy as np

from . import Simulation
from .. import _runtime

logger = logging.getLogger(__name__)


class ScalarFieldSimulation(Simulation):
    def __init__(self, sensor, cluster_radius: float = None):
        super().__init__(sensor)
        self.cluster_radius = cluster_radius

    def _engine(self):
        gm = self.sensor.grid_map
        mp = gm.params.get("mapping", {})
        cfg = _runtime.engine_config_from(gm, self.sensor, mp.get("signal_variance", 1.0), mp.get("length_scale", 1.0),
                                          cluster_radius=self.cluster_radius)
        return _runtime.compat_engine(cfg)

    def create_ground_truth_map(self) -> np.array:
        raise NotImplementedError("Scalar field simulation has no function implemented to create ground truth map")

    def take_measurement(self, position: np.array, verbose: bool = True) -> np.array:
        """crop -> downsample -> + noise -> clip (reference :26-34), on the device."""
        eng = self._engine()
        eng.write_gt(1, self.ground_truth_map)
        xl, xr, yu, yd = self.sensor.project_field_of_view(position)
        rf = self.sensor.get_resolution_factor(position)
        h, w = yd - yu + 1, xr - xl + 1
        shape = (h, w) if rf == 1 else (math.ceil(w / rf), math.ceil(h / rf))  # cv2 dsize transposition, SURVEY a17
        eps = np.random.normal(0, 1, shape)  # legacy stream; scaled by the noise variance on the device
        z, m, shp = eng.observe(np.asarray(position, dtype=np.float64).reshape(1, 3), env_ids=[1], meas_noise=eps.reshape(1, -1))
        m = int(m[0])
        if m != eps.size or tuple(int(v) for v in shp[0]) != shape:
            logger.error(f"observation shape mismatch: device {tuple(int(v) for v in shp[0])} vs host {shape}")
            raise ValueError
        return _runtime.to_host64(z[0, :m]).reshape(shape)


class GaussianRandomField(ScalarFieldSimulation):
    def __init__(self, sensor, cluster_radius: float):
        super().__init__(sensor, cluster_radius)
        self.ground_truth_map = self.create_ground_truth_map()

    def create_ground_truth_map(self) -> np.array:
        """2-D Gaussian random field in [0, 1] (reference :43-47 -> simulations/ground_truths.py:14-33)."""
        gm = self.sensor.grid_map
        if gm.x_dim != gm.y_dim:
            logger.error("The reference swaps x_dim / y_dim here (simulations.py:45-47); only square grids are defined")
            raise ValueError
        white = np.random.normal(size=(gm.x_dim, gm.y_dim))  # one draw from the legacy stream, like the reference
        eng = self._engine()
        eng.reset(env_ids=[1], white_noise=white[None])
        return _runtime.to_host64(eng.read_gt(1))
