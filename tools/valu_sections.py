"""Dynamic instruction counts of k_step_patch by section: an -DIPP_EXIT_POINTS=1 build returns at a chosen point
(ipp_debug_capture(1 + point)), and rocprofv3's per-dispatch SQ_INSTS_* of that one launch is the count up to there.
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d out -o p -- \
        python3 tools/valu_sections.py <point>          (IPP_HIP_LIB = the exit-point build; see tools/valu_sections.sh)
points: 1 header, 2 rectangle tests + tables, 3 gather + records, 4 m x m algebra / observation, 5 units, 0 whole kernel;
6 .. 9: the whole kernel with one section of the unit loop skipped (prior term, row stream, L^-1, stores)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

point = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, T = 4096, 40
cfg = EngineConfig(x_dim=50, y_dim=50)
alts = [float(a) for a in range(5, 15)]
env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, parts=1)
env.reset()
for t in range(T + 6):
    env.step(cell_centre_actions(cfg, t, 0, B, B, alts))
torch.cuda.synchronize()
if point:
    env.engine._lib.ipp_debug_capture(env.engine._h, 1 + point)
env.step(cell_centre_actions(cfg, T + 6, 0, B, B, alts))  # the LAST dispatch of k_step_patch is the one to read
torch.cuda.synchronize()
