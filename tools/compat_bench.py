import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.params import example_params
from ipp_rl_amd.mapping.grid_maps import GridMap
from ipp_rl_amd.mapping.mappings import Mapping
from ipp_rl_amd.sensors.cameras import RGBCamera
from ipp_rl_amd.sensors.models.sensor_models import AltitudeSensorModel
from ipp_rl_amd.simulations.simulations import GaussianRandomField
from ipp_rl_amd.planning.common.optimization import simulate_prediction_step
params = example_params(50)
np.random.seed(0)
gm = GridMap(params); sensor = RGBCamera(params["sensor"]["field_of_view"], AltitudeSensorModel(0.05, 0.2), gm)
sim = GaussianRandomField(sensor, 5); sensor.set_sensor_simulation(sim)
mapping = Mapping(gm, sensor)
uav = {"max_v": 2, "max_a": 2}
info = {"mean": gm.mean, "value_threshold": 0.4, "interval_factor": 0}
prev = np.array([2., 2., 14.]); P = gm.cov_matrix
rs = np.random.RandomState(1)
acts = [np.array([4.*rs.randint(50)+2, 4.*rs.randint(50)+2, float(rs.randint(5,15))]) for _ in range(30)]
simulate_prediction_step(P, prev, acts[0], mapping, uav, info)
t0 = time.perf_counter()
for a in acts[:20]:
    r, _, Pn = simulate_prediction_step(P, prev, a, mapping, uav, info)
dt = (time.perf_counter() - t0) / 20
print("simulate_prediction_step (drop-in, 50x50): %.2f ms/call = %.0f calls/s" % (dt*1e3, 1/dt))
# chained, the way the tree searches use it (mcts.py:239, mcts_mission.py:228-246): next_state feeds the next call, the
# planner reads np.diag(state) of every node
Pc, pv = P, prev
t0 = time.perf_counter()
for a in acts[:20]:
    r, _, Pc = simulate_prediction_step(Pc, pv, a, mapping, uav, info)
    d = np.diag(Pc)
    pv = a
dt = (time.perf_counter() - t0) / 20
print("chained simulate_prediction_step + np.diag(next_state): %.2f ms/call" % (dt*1e3))
t0 = time.perf_counter(); full = np.asarray(Pc); dt = time.perf_counter() - t0
print("materialising one state on the host (np.asarray): %.1f ms, trace %.4f vs diag sum %.4f" % (dt*1e3, np.trace(full), d.sum()))
t0 = time.perf_counter()
for a in acts[:10]:
    z = sensor.take_measurement(a, verbose=False); mapping.update_grid_map(a, z)
dt = (time.perf_counter() - t0) / 10
print("take_measurement + update_grid_map (drop-in): %.2f ms/step" % (dt*1e3))
