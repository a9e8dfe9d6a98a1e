#!/usr/bin/env python3
"""BASELINE configs[4] through the tree-search drivers on their own (rocprofv3 passes, A/B runs).
usage: python tools/mcts_bench.py [--driver device|host] [--roots 1024] [--sims 256] [--grid 200] [--reps 2] [--in-flight 4]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--driver", default="device")
ap.add_argument("--roots", type=int, default=1024)
ap.add_argument("--sims", type=int, default=256)
ap.add_argument("--grid", type=int, default=200)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--in-flight", type=int, default=4)
a = ap.parse_args()
for _ in range(a.reps):
    r = bench.run_mcts_driver(torch, torch.device("cuda:0"), grid=a.grid, roots=a.roots, sims=a.sims, driver=a.driver, in_flight=a.in_flight)
    print(json.dumps({k: r[k] for k in ("value", "seconds_per_search", "seconds_per_search_device_policies", "device_tree_steps", "launches", "nodes", "all_policies_valid")}))
