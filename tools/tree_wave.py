#!/usr/bin/env python3
"""The configs[4] tree wave of bench.py on its own (for rocprofv3 passes and A/B runs of ipp_tree_step).
usage: python tools/tree_wave.py [--wave 4] [--reps 2] [--roots 1024] [--grid 200]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--wave", type=int, default=4)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--roots", type=int, default=1024)
ap.add_argument("--grid", type=int, default=200)
a = ap.parse_args()
r = bench.run_tree_wave(torch, torch.device("cuda:0"), grid=a.grid, roots=a.roots, reps=a.reps, wave=a.wave)
print(json.dumps({k: r[k] for k in ("value", "kernel", "kernel_ms_avg", "achieved_gbs", "frac", "launch_items", "all_status_ok")}))
