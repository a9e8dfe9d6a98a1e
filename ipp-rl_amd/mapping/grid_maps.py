"""GridMap: environment dimensions + the belief arrays callers read and assign (reference mapping/grid_maps.py:7-54)."""
import logging
from typing import Dict

logger = logging.getLogger(__name__)


def _require(params: Dict, section: str, key: str, what: str):
    """Config lookup with the reference's error behaviour: log an error, raise a bare ValueError."""
    if section not in params.keys():
        logger.error(f"Cannot find {section} specification in config file!")
        raise ValueError
    if key not in params[section].keys():
        logger.error(f"Cannot find {what} specification in config file!")
        raise ValueError
    return params[section][key]


class GridMap:
    def __init__(self, params: Dict):
        self.params = params
        self.mean = None        # (y_dim, x_dim) float64
        self.cov_matrix = None  # (N, N) float64

    @property
    def x_dim(self) -> int:
        """map width in cells"""
        return _require(self.params, "environment", "x_dim", "environment's x_dim")

    @property
    def y_dim(self) -> int:
        """map height in cells"""
        return _require(self.params, "environment", "y_dim", "environment's y_dim")

    @property
    def resolution(self):
        """metres per cell"""
        return _require(self.params, "environment", "resolution", "environment's resolution")

    @property
    def num_grid_cells(self) -> int:
        return self.x_dim * self.y_dim
