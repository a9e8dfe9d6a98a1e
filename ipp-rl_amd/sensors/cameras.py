"""
Camera / RGBCamera (reference sensors/cameras.py:13-125): footprint geometry on the host in fp64 (the HIP
prologue kernel repeats the same fp64 arithmetic per step on the device); take_measurement delegates to the
simulation, whose crop / downsample / noise / clip runs on the device (ipp_observe).
"""
import logging
from typing import Dict, Tuple

import numpy as np

from . import Sensor

logger = logging.getLogger(__name__)


class Camera(Sensor):
    def __init__(self, field_of_view: Dict, sensor_model, grid_map):
        super().__init__(sensor_model, grid_map)
        self.field_of_view = field_of_view

    @property
    def angle_x(self) -> float:
        return self.field_of_view["angle_x"]

    @property
    def angle_y(self) -> float:
        return self.field_of_view["angle_y"]

    def field_of_view_range(self, height: float) -> Tuple[float, float]:
        """Ground-plane extent [m] of the frustum at `height` (reference :34-47)."""
        return (2 * height * np.tan(0.5 * np.radians(self.angle_x)),
                2 * height * np.tan(0.5 * np.radians(self.angle_y)))

    def project_field_of_view(self, position: np.array) -> Tuple[int, int, int, int]:
        """Clipped cell rectangle (xl, xr, yu, yd) seen from `position` (reference :49-75)."""
        res = self.grid_map.resolution
        ext_x, ext_y = self.field_of_view_range(position[2])
        rad = np.floor(0.5 * np.floor(np.array([ext_x, ext_y]) / res))
        centre = np.floor(np.asarray(position[:2], dtype=np.float64) / res)
        lo, hi = centre - rad, centre + rad
        xl, xr = np.clip([lo[0], hi[0]], 0, self.grid_map.x_dim - 1)
        yu, yd = np.clip([lo[1], hi[1]], 0, self.grid_map.y_dim - 1)
        return int(xl), int(xr), int(yu), int(yd)

    def take_measurement(self, position: np.array, verbose: bool = True) -> np.array:
        pass

    def process_measurement(self, image: np.array) -> np.array:
        pass

    def get_resolution_factor(self, position: np.array) -> float:
        pass


class RGBCamera(Camera):
    def __init__(self, field_of_view: Dict, sensor_model, grid_map, encoding: str = "rgb8"):
        super().__init__(field_of_view, sensor_model, grid_map)
        self.encoding = encoding

    def take_measurement(self, position: np.array, verbose: bool = True) -> np.array:
        """Simulated measurement, or a random RGB image when no simulation is attached (reference :108-116)."""
        if verbose:
            logger.info(f"Take measurement at point: {position}")
        if self.sensor_simulation is None:
            return (np.random.random((self.grid_map.x_dim, self.grid_map.y_dim, 3)) * 255).astype(int)
        return self.sensor_simulation.take_measurement(position)

    def process_measurement(self, image: np.array) -> np.array:
        return image

    def get_resolution_factor(self, position: np.array) -> float:
        """2 strictly above 10 m (reference :122-125)."""
        return 2 if position[2] > 10.0 else 1
