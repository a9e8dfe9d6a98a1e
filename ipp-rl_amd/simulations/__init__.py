"""Simulation base class (reference simulations/__init__.py:11-31; the matplotlib viewer is not part of the path)."""
import numpy as np


class Simulation:
    def __init__(self, sensor):
        super(Simulation, self).__init__()
        self.sensor = sensor
        self.ground_truth_map = None

    def create_ground_truth_map(self):
        raise NotImplementedError("Sensor simulation has no function implemented to create ground truth map")

    def take_measurement(self, position: np.array, verbose: bool = True):
        raise NotImplementedError("Sensor simulation has no function implemented to take measurement")

    def get_ground_truth_submap(self, xl: int, xr: int, yu: int, yd: int) -> np.array:
        return self.ground_truth_map[yu : yd + 1, xl : xr + 1]
