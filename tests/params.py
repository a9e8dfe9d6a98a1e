"""Hot-path values of the reference's config/example.yaml as plain data."""
import copy

EXAMPLE = {
    "environment": {"x_dim": 10, "y_dim": 10, "resolution": 4},
    "sensor": {
        "type": "rgb_camera",
        "field_of_view": {"angle_x": 60, "angle_y": 60},
        "encoding": "rgb8",
        "model": {"type": "altitude_dependent", "coeff_a": 0.05, "coeff_b": 0.2},
        "simulation": {"type": "gaussian_random_field", "cluster_radius": 5},
    },
    "mapping": {"fit_gaussian_process": True, "prior_cov_mean": 0.5, "prior_cov_std": 0.25, "signal_variance": 1.82,
                "length_scale": 3.67, "noise_variance": 1.42, "nu": 1.5},
    "experiment": {
        "constraints": {"min_altitude": 8, "max_altitude": 14, "altitude_spacing": 6, "budget": 200},
        "scenario": {"adaptive": True, "value_threshold": 0.4, "interval_factor": 0},
        "uav": {"max_v": 2, "max_a": 2, "sampling_time": 2},
    },
}


def example_params(x_dim=10, y_dim=None, resolution=4):
    p = copy.deepcopy(EXAMPLE)
    p["environment"].update(x_dim=x_dim, y_dim=y_dim or x_dim, resolution=resolution)
    return p
