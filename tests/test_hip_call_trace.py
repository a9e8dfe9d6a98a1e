"""
Replay of the reference planners' CALL TRACE against this repo's drop-in classes (tests/golden/call_trace.npz, recorded by
tests/golden/gen_golden.py::gen_call_trace with recording proxies around the reference's Mapping / GridMap / sensor / simulation):
  mission   two rounds of MCTSMission's loop -- replan (run_simulations_proxy + select_best_child) and the executed step
            (/root/reference planning/mcts_mission.py:352-413, eval: planning/missions.py:176-197)
  selfplay  two steps of the self-play episode loop (planning/mcts_zero/episode_generators.py:112-155) with the reference's MCTS on
            stubbed inference queues
Every event is an attribute read or a call the planner made on the class surface (or on the planning.common helpers it imports by
name), with its arguments and its result.  The replay issues the same sequence on Mapping / GridMap / RGBCamera /
GaussianRandomField of ipp_rl_amd -- an argument that was the RESULT of an earlier event is OUR result of that event (a DeviceCov
state goes back in the way the planner hands it back) -- and compares every result within 1e-5.  The reference's planners are not
needed on the GPU box: the trace is data.
"""
import numpy as np
import pytest

from tests.params import example_params
from tests.trace_replay import TOL, run_trace

pytestmark = pytest.mark.gpu


def _build(dim, seed):
    from ipp_rl_amd.mapping.grid_maps import GridMap
    from ipp_rl_amd.mapping.mappings import Mapping
    from ipp_rl_amd.sensors.cameras import RGBCamera
    from ipp_rl_amd.sensors.models.sensor_models import AltitudeSensorModel
    from ipp_rl_amd.simulations.simulations import GaussianRandomField

    params = example_params(dim)
    np.random.seed(seed)
    gm = GridMap(params)
    sm = AltitudeSensorModel(0.05, 0.2)
    sensor = RGBCamera(params["sensor"]["field_of_view"], sm, gm)
    sim = GaussianRandomField(sensor, 5)
    sensor.set_sensor_simulation(sim)
    return gm, sensor, sim, Mapping(gm, sensor, shuffle_prior_cov=False)


def _run(golden, prefix):
    from ipp_rl_amd.planning.common import actions as our_actions
    from ipp_rl_amd.planning.common import rewards as our_rewards
    from ipp_rl_amd.planning.common.optimization import simulate_prediction_step

    g = golden("call_trace")
    gm, sensor, sim, mapping = _build(10, int(g[prefix + "_seed"]))
    objs = {"mapping": mapping, "mapping.grid_map": gm, "mapping.sensor": sensor, "mapping.sensor.sensor_simulation": sim,
            "mapping.sensor.sensor_model": sensor.sensor_model}
    fns = {"compute_reward": our_rewards.compute_reward, "compute_adaptive_msk": our_rewards.compute_adaptive_msk,
           "action_costs": our_actions.action_costs}

    def simulate(state, prev, action, uav, info):
        reward, _, next_state = simulate_prediction_step(state, prev, action, mapping, uav, info)
        return reward, next_state

    rp, kinds = run_trace(g, prefix, objs, fns, simulate)
    return rp, kinds, (gm, sensor, sim, mapping)


def test_mcts_mission_call_trace_replays_on_the_drop_in_classes(golden):
    rp, kinds, (gm, sensor, sim, mapping) = _run(golden, "mission")
    # what the mission loop does: predict steps on states it got back, rewards on them, two executed measurements
    assert kinds[("call", "mapping", "update_grid_map")] >= 40 and kinds[("call", "mapping.sensor", "take_measurement")] == 2
    assert kinds[("call", "fn", "compute_reward")] >= 40 and kinds[("get", "mapping.grid_map", "cov_matrix")] >= 3
    assert rp.checked > 150 and rp.worst < TOL
    # the metrics eval() derives from the attributes it read (planning/missions.py:176-197), on our final map
    from oracle import ipp_oracle as orc

    gt = sim.ground_truth_map
    msk = gt.flatten(order="C") >= 0.4
    want = golden("call_trace")["mission_metrics"]
    assert abs(orc.metric_rmse(gt, gm.mean, msk) - want[0][-1]) < TOL
    assert abs(orc.metric_uncertainty(np.diag(np.asarray(gm.cov_matrix)), msk) - want[4][-1]) < 1e-4


def test_self_play_call_trace_replays_on_the_drop_in_classes(golden):
    rp, kinds, (gm, sensor, sim, mapping) = _run(golden, "selfplay")
    assert kinds[("call", "mapping", "update_grid_map")] >= 40 and kinds[("call", "mapping.sensor", "take_measurement")] == 2
    assert kinds[("call", "fn", "simulate_prediction_step")] == 2
    assert rp.checked > 150 and rp.worst < TOL
    # a state the planner holds is usable the ways the reference's MCTS uses it (mcts.py:20-21 node key, features.py diag)
    state = gm.cov_matrix
    assert str(state) == str(np.asarray(state)) and np.allclose(np.diag(state), np.diag(np.asarray(state)))
