#!/usr/bin/env python3
"""
Time of the device GRF generator (ipp_generate_grf: white noise -> normalised field, simulations/ground_truths.py:14-33)
for the batch sizes the bench configs reset per step.

usage: python tools/grf_bench.py [grid:fields ...]      default 50:102 100:2048 200:102
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ipp_rl_amd import EngineConfig, IPPEngine  # noqa: E402


def main():
    specs = sys.argv[1:] or ["50:102", "100:2048", "200:102"]
    for spec in specs:
        n, fields = (int(x) for x in spec.split(":"))
        eng = IPPEngine(EngineConfig(x_dim=n, y_dim=n), capacity=fields, state="factor", rank_cap=16, window_rows=0)
        white = torch.randn((fields, n * n), dtype=torch.float32, device="cuda")
        out = torch.empty_like(white)
        for _ in range(3):
            eng.generate_grf(white, out=out)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.generate_grf(white, out=out)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        fma = 3.0 * n ** 3 * fields  # half-spectrum DFT form (k_grf_dft.h)
        npad = 16 * ((n + 15) // 16)
        gemm = 4.0 * npad ** 3 * fields if n % 2 == 0 and n <= 128 else 0.0  # four padded GEMMs (k_grf_hartley.h)
        print(f"[{n}x{n}, {fields} fields] {ms:.3f} ms per call, {fields / ms * 1e3:.3e} fields/s, "
              f"{2 * fma / ms / 1e9:.1f} fp64 TFLOP/s on the 3 n^3 FMA count"
              + (f", {2 * gemm / ms / 1e9:.1f} TFLOP/s on the padded GEMMs" if gemm else ""))
        # the same with the white noise drawn inside the generator (what the batched env driver runs)
        if eng.generate_grf_rows(fields, 3, 1 << 40, out):
            for _ in range(3):
                eng.generate_grf_rows(fields, 3, 1 << 40, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.generate_grf_rows(fields, 3, 1 << 40, out)
            torch.cuda.synchronize()
            print(f"    noise drawn in the kernel: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call")
        eng.close()


if __name__ == "__main__":
    main()
