"""
AltitudeSensorModel (reference sensors/models/sensor_models.py:14-85): host-side scalar helpers.  The HIP
prologue kernel (csrc/k_prepare.h) evaluates the same quantities per step on the device and never builds
the dense m x N matrix; measurement_model_matrix is kept for callers that want it explicitly.
"""
import numpy as np

from . import SensorModel


class AltitudeSensorModel(SensorModel):
    def __init__(self, coeff_a: float, coeff_b: float):
        super().__init__()
        self.coeff_a = coeff_a
        self.coeff_b = coeff_b

    def get_noise_variance(self, position: np.array) -> float:
        """a (1 - exp(-b altitude))  (reference :27-30)"""
        return self.coeff_a * (1 - np.exp(-self.coeff_b * position[2]))

    def measurement_variance_matrix(self, position: np.array, num_measurements: int, resolution_factor: float):
        """rf^3 * noise * I_m  (reference :32-36)"""
        return resolution_factor ** 3 * self.get_noise_variance(position) * np.identity(num_measurements)

    @staticmethod
    def footprint_blocks(field_of_view_indices, resolution_factor: int, x_dim: int):
        """Yield (flat cell indices, weight) per measurement row: rf x rf blocks clipped to the footprint,
        weight 1/rf^2 for complete blocks and 1/rf otherwise (reference :57-79)."""
        xl, xr, yu, yd = field_of_view_indices
        rf = int(resolution_factor)
        w, h = xr - xl + 1, yd - yu + 1
        nx, ny = (w - 1) // rf + 1, (h - 1) // rf + 1
        for i in range(nx * ny):
            by, bx = divmod(i, nx)
            xs = range(bx * rf, min(bx * rf + rf, w))
            ys = range(by * rf, min(by * rf + rf, h))
            cells = np.array([x_dim * (yu + ly) + xl + lx for ly in ys for lx in xs], dtype=np.int64)
            yield cells, (1 / rf ** 2 if len(cells) >= rf ** 2 else 1 / rf)

    def measurement_model_matrix(self, grid_map, field_of_view_indices, num_measurements, resolution_factor: int):
        """Dense H (m x N), rows = block averages over the footprint (reference :38-81)."""
        H = np.zeros((int(num_measurements), grid_map.num_grid_cells))
        for i, (cells, weight) in enumerate(self.footprint_blocks(field_of_view_indices, resolution_factor, grid_map.x_dim)):
            if i < H.shape[0]:
                H[i, cells] = weight
        return H

    @staticmethod
    def flatten_2d_indices(indices_2d: np.array, x_dim: int) -> np.array:
        return (x_dim * indices_2d[:, 0] + indices_2d[:, 1]).astype(int)
