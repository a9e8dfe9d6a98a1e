"""Fast / slow placement modes of a large arena: is it the memory system or the kernel, and can it be selected?
One process, configs[3] share by default (32768 envs of 50x50).  For every arena: the BARE row-stream probe (ipp_arena_probe: no
kernel logic) and the REAL two-group step on the same memory.
  phase A  arenas one after the other (allocated, measured, freed): torch, hipMalloc, VMM with 2-MiB / 64-MiB / 1-GiB chunks
  phase B  K hipMalloc arenas held AT ONCE (distinct physical memory by construction), each measured
usage: python tools/arena_modes.py [envs] [grid] [steps] [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ipp_rl_amd import EngineConfig
from ipp_rl_amd.engine import DeviceArena
from ipp_rl_amd.vec_env import VecIPPEnv, cell_centre_actions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
G = int(sys.argv[2]) if len(sys.argv) > 2 else 50
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
T = 40 if G == 50 else 16
cfg = EngineConfig(x_dim=G, y_dim=G)
ALTS = [float(a) for a in range(5, 15)]
acts = [torch.as_tensor(cell_centre_actions(cfg, t, 0, B, B, ALTS), device="cuda") for t in range(T + 4 * steps)]


def real(arena):
    env = VecIPPEnv(cfg, B, episode_steps=T, stagger=True, window_rows=-1, seed=1, parts=2, arena=arena)
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
    nbytes = env.engine.arena_bytes
    ptr = env.engine.arena.data_ptr()
    env.close(); del env
    return times, nbytes, ptr


# size of the arena: from a throw-away engine on a torch tensor (phase A's first line)
def line(tag, arena, ptr=None, nbytes=None):
    pr = lat = None
    if isinstance(arena, DeviceArena):
        pr = [arena.probe(B, rows=32, launches=5) for _ in range(2)]
        lat = [arena.latency(256, 2000), arena.latency(256, 2000), arena.latency(8192, 500), arena.latency(8192, 500)]
    times, nb, p = real(arena)
    gb = B * 32 * 656 * 4 / 1e9
    prs = "-" if pr is None else " ".join(f"{x:.4f}" for x in pr) + f" ms ({gb / (1e-3 * min(pr)) / 1e3:.2f} TB/s)"
    las = "-" if lat is None else " ".join(f"{x:.0f}" for x in lat)
    print(f"{tag:34s} at 0x{p:x} {nb / 2**30:6.1f} GiB  step ms {[round(x, 4) for x in times]}  probe {prs}  latency ns/hop (256 waves x2, 8192 waves x2) {las}", flush=True)
    return min(times), (lat[0] if lat else None), nb


print(f"envs {B}, grid {G}x{G}, {steps} steps per region, T {T}", flush=True)
_, _, NB = line("A torch.empty", "torch")
torch.cuda.empty_cache()
pairs = []
for rep in range(2):
    for kind, chunk, align in (("hip", 0, 0), ("vmm", 2 << 20, 2 << 20), ("vmm", 2 << 20, 1 << 30), ("vmm", 64 << 20, 64 << 20), ("vmm", 1 << 30, 1 << 30),
                               ("vmm", 1 << 30, 2 << 20), ("vmm", 2 << 30, 2 << 30), ("hip", 0, 0)):
        try:
            a = DeviceArena(NB, 0, kind=kind, chunk_bytes=chunk, align_bytes=align)
        except Exception as e:
            print(f"A {kind} chunk {chunk >> 20} MiB: {e}", flush=True)
            continue
        st, pr, _ = line(f"A {kind} chunk {chunk >> 20} align {align >> 20} MiB rep {rep}", a)
        pairs.append((st, pr))
        a.free()
print("phase B: arenas held at once", flush=True)
held = []
for k in range(K):
    try:
        held.append(DeviceArena(NB, 0, kind="hip"))
    except Exception as e:
        print(f"B arena {k}: {e}", flush=True)
        break
for rep in range(2):
    for k, a in enumerate(held):
        st, pr, _ = line(f"B held {k} of {len(held)} rep {rep}", a)
        pairs.append((st, pr))
for a in held:
    a.free()
import numpy as np
x = np.array(pairs)
if len(x) > 2:
    print(f"correlation(step ms, idle latency ns) over {len(x)} arenas: {np.corrcoef(x[:, 0], x[:, 1])[0, 1]:.3f};  step min {x[:, 0].min():.4f} max {x[:, 0].max():.4f};  probe min {x[:, 1].min():.4f} max {x[:, 1].max():.4f}")
