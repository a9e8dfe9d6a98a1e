"""Does the step time depend on WHERE the arena lies?  configs[3] share (32768 envs, 50x50) / configs[1], same process, a dummy
allocation of varying size in front of the env's arena.  usage: python tools/arena_shift.py [envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = 40
cfg = EngineConfig(x_dim=50, y_dim=50)
ALTS = [float(a) for a in range(5, 15)]
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(T + 4 * steps)]
for shift_mb in [0] * int(os.environ.get("REPS", "8")):
    pad = torch.empty(shift_mb << 20, dtype=torch.uint8, device="cuda") if shift_mb else None
    env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=2)
    env.reset()
    t = 0
    for _ in range(T + steps):
        env.step_async(acts[t], inputs_ready=True); t += 1
    env.wait(); torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            env.step_async(acts[T + steps + i], inputs_ready=True)
        env.wait(); torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0) / steps)
    qr = env.queue_report or {}
    import subprocess
    clk = subprocess.run("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | head -2 | tr -s ' ' | tr '\\n' ' '", shell=True, capture_output=True, text=True).stdout.strip()
    print(f"pad {shift_mb:5d} MiB  arena at 0x{env.engine.arena.data_ptr():x}  ms per step {[round(x, 4) for x in times]}  queues {qr.get('n_queues')} distinct {qr.get('parts_distinct')} staging_shares {qr.get('staging_shares_a_part_queue')} on_caller {qr.get('staging_on_callers_queue')} probed {qr.get('streams_probed')} | {clk}")
    env.close(); del env, pad
    if os.environ.get("KEEP_CACHE") != "1":
        torch.cuda.empty_cache()
