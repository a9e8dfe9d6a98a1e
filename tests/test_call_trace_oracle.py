"""
The call-trace fixture (tests/golden/call_trace.npz: what the reference's planners do to the class surface, recorded from the
imported reference) replayed on ORACLE-backed stand-ins on the CPU: pins the fixture and the replay engine to the golden-pinned
restatement, so that the GPU replay (tests/test_hip_call_trace.py) compares the drop-in classes with something this suite has checked.
mapping/mappings.py:114-153, sensors/cameras.py:76-85, planning/common/{rewards,actions,optimization}.py.
"""
import numpy as np

from oracle import ipp_oracle as orc
from tests.trace_replay import TOL, run_trace


class _Grid:
    def __init__(self, cfg):
        self.cfg, self.resolution, self.x_dim, self.y_dim = cfg, cfg.resolution, cfg.x_dim, cfg.y_dim
        self.mean = 0.5 * np.ones((cfg.y_dim, cfg.x_dim))
        self.cov_matrix = orc.matern_prior(cfg)


class _Sim:
    def __init__(self, gt):
        self.ground_truth_map = gt


class _Sensor:
    def __init__(self, cfg, sim):
        self.cfg, self.sensor_simulation = cfg, sim

    def take_measurement(self, position, verbose=True):
        fov = orc.project_fov(self.cfg, position)
        rf = orc.resolution_factor(position)
        ny, nx = -(-(fov[3] - fov[2] + 1) // rf), -(-(fov[1] - fov[0] + 1) // rf)
        eps = np.random.normal(0, 1, (nx, ny) if rf > 1 else (ny, nx))  # (the reference's transposed dsize, sensor_manipulations.py:22)
        return orc.observe(self.cfg, self.sensor_simulation.ground_truth_map, position, eps)


class _Mapping:
    def __init__(self, cfg, gm, sensor):
        self.cfg, self.grid_map, self.sensor = cfg, gm, sensor

    def update_grid_map(self, position, measurement=None, cov_only=False, predict_only=False, current_cov_matrix=None):
        P = self.grid_map.cov_matrix if current_cov_matrix is None else current_cov_matrix
        x, P_new, _ = orc.update_grid_map(self.cfg, P, self.grid_map.mean, position, z=measurement, cov_only=cov_only)
        if predict_only:
            return x, P_new
        self.grid_map.cov_matrix = P_new  # (the reference returns nothing here, mappings.py:152-153)
        if x is not None:
            self.grid_map.mean = x


def _objects(seed):
    cfg = orc.OracleConfig(x_dim=10, y_dim=10)
    np.random.seed(seed)
    gt = orc.gaussian_random_field(cfg, rng=np.random)
    gm = _Grid(cfg)
    sim = _Sim(gt)
    sensor = _Sensor(cfg, sim)
    mapping = _Mapping(cfg, gm, sensor)
    objs = {"mapping": mapping, "mapping.grid_map": gm, "mapping.sensor": sensor, "mapping.sensor.sensor_simulation": sim}
    fns = {
        "compute_reward": lambda cur, nxt, prev, act, uav=None, msk=None: orc.reward_from_diags(np.diag(cur), np.diag(nxt), act, prev, uav, msk),
        "compute_adaptive_msk": lambda mean, cov, thr, kf: orc.adaptive_mask(mean, cov, thr, kf),
        "action_costs": lambda act, prev, uav=None: orc.action_cost(act, prev, uav),
    }

    def simulate(state, prev, action, uav, info):
        reward, P_new, _, _ = orc.predict_step(cfg, state, prev, action, uav, info)
        return reward, P_new

    return cfg, objs, fns, simulate


def test_call_trace_replays_on_the_oracle(golden):
    g = golden("call_trace")
    for prefix in ("mission", "selfplay"):
        cfg, objs, fns, simulate = _objects(int(g[prefix + "_seed"]))
        rp, kinds = run_trace(g, prefix, objs, fns, simulate)
        assert kinds[("call", "mapping", "update_grid_map")] >= 40 and kinds[("call", "mapping.sensor", "take_measurement")] == 2
        assert rp.checked > 150 and rp.worst < 1e-9, (prefix, rp.worst)
