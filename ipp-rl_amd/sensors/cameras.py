"""
``Camera`` / ``RGBCamera`` at the reference's module path (sensors/cameras.py).  The footprint geometry is evaluated
on the host in fp64 for callers that ask for it; inside an env step the HIP prologue repeats the same fp64
arithmetic on the device, and ``take_measurement`` ends in the device-side crop / downsample / noise / clip
(``ipp_observe``) through the attached simulation.
"""
import logging
import math
from typing import Dict, Tuple

import numpy as np

from . import Sensor

logger = logging.getLogger(__name__)

ALTITUDE_OF_HALF_RESOLUTION = 10.0  # metres: strictly above it the camera reports every 2 x 2 block as one pixel


def _ground_extent(height: float, opening_angle_deg: float) -> float:
    """Width [m] of the strip a pinhole with the given full opening angle sees from `height`."""
    return 2 * height * np.tan(0.5 * np.radians(opening_angle_deg))


def _clipped_span(centre_cell: float, radius_cells: float, n_cells: int) -> Tuple[int, int]:
    first, last = np.clip([centre_cell - radius_cells, centre_cell + radius_cells], 0, n_cells - 1)
    return int(first), int(last)


class Camera(Sensor):
    """Downward-looking pinhole camera with a rectangular field of view given by two opening angles [deg]."""

    def __init__(self, field_of_view: Dict, sensor_model, grid_map):
        Sensor.__init__(self, sensor_model, grid_map)
        self.field_of_view = field_of_view

    angle_x = property(lambda self: self.field_of_view["angle_x"])
    angle_y = property(lambda self: self.field_of_view["angle_y"])

    def field_of_view_range(self, height: float) -> Tuple[float, float]:
        return _ground_extent(height, self.angle_x), _ground_extent(height, self.angle_y)

    def project_field_of_view(self, position: np.array) -> Tuple[int, int, int, int]:
        """(xl, xr, yu, yd): the cell rectangle under the camera, clipped to the map.  The radius is half the number
        of whole cells the ground extent spans, rounded down, around the cell that contains the position."""
        cell = self.grid_map.resolution
        extent_x, extent_y = self.field_of_view_range(position[2])
        radius_x, radius_y = (math.floor(0.5 * math.floor(extent / cell)) for extent in (extent_x, extent_y))
        col, row = (math.floor(float(coordinate) / cell) for coordinate in position[:2])
        xl, xr = _clipped_span(col, radius_x, self.grid_map.x_dim)
        yu, yd = _clipped_span(row, radius_y, self.grid_map.y_dim)
        return xl, xr, yu, yd

    # hooks a concrete camera fills in (the reference leaves them empty here rather than raising)
    def take_measurement(self, position: np.array, verbose: bool = True) -> np.array:
        return None

    def process_measurement(self, image: np.array) -> np.array:
        return None

    def get_resolution_factor(self, position: np.array) -> float:
        return None


class RGBCamera(Camera):
    def __init__(self, field_of_view: Dict, sensor_model, grid_map, encoding: str = "rgb8"):
        Camera.__init__(self, field_of_view, sensor_model, grid_map)
        self.encoding = encoding

    def get_resolution_factor(self, position: np.array) -> float:
        return 2 if position[2] > ALTITUDE_OF_HALF_RESOLUTION else 1

    def process_measurement(self, image: np.array) -> np.array:
        return image

    def take_measurement(self, position: np.array, verbose: bool = True) -> np.array:
        if verbose:
            logger.info(f"Take measurement at point: {position}")
        simulation = self.sensor_simulation
        if simulation is not None:
            return simulation.take_measurement(position)
        # no simulation attached: the reference hands back a random 8-bit RGB image of the map's shape
        shape = (self.grid_map.x_dim, self.grid_map.y_dim, 3)
        return (255 * np.random.random(shape)).astype(int)
