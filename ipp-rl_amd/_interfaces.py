"""
The three abstract bases of the reference's class surface -- ``SensorModel``, ``Sensor``, ``Simulation``
(reference sensors/models/__init__.py, sensors/__init__.py, simulations/__init__.py) -- kept in one place.
Concrete subclasses live in the mirrored modules; anything a subclass leaves out raises ``NotImplementedError``
like the reference's stubs do.
"""
from __future__ import annotations


def _unimplemented(what: str):
    """A method that reports which hook of which concrete class is missing."""

    def hook(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}: no {what} implemented")

    hook.__doc__ = f"Subclass hook: {what}."
    return hook


class SensorModel:
    """Noise characteristics of a sensor as a function of the UAV position."""

    get_noise_variance = _unimplemented("noise variance function")


class Sensor:
    """A sensor bound to its noise model and to the grid map it observes; a simulation may be attached later."""

    def __init__(self, sensor_model, grid_map):
        self.sensor_model, self.grid_map = sensor_model, grid_map
        self.sensor_simulation = None

    def set_sensor_simulation(self, sensor_simulation) -> None:
        self.sensor_simulation = sensor_simulation

    take_measurement = _unimplemented("measuring function")
    process_measurement = _unimplemented("processing function")
    get_resolution_factor = _unimplemented("resolution factor function")


class Simulation:
    """Owner of the ground-truth map a simulated sensor crops its measurements from."""

    def __init__(self, sensor):
        self.sensor = sensor
        self.ground_truth_map = None

    create_ground_truth_map = _unimplemented("ground truth generator")
    take_measurement = _unimplemented("measurement function")

    def get_ground_truth_submap(self, xl: int, xr: int, yu: int, yd: int):
        """Inclusive footprint rectangle of the ground truth (rows yu..yd, columns xl..xr)."""
        return self.ground_truth_map[yu:yd + 1, xl:xr + 1]
