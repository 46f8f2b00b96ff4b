#!/usr/bin/env python3
"""bench.py's Chamfer and pose-search legs alone (no CPU baselines)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import bench_legs     # noqa: E402

dev = torch.device("cuda:0")
print(json.dumps(bench_legs.chamfer_leg(dev, cpu=False)))
print(json.dumps(bench_legs.pose_search_leg(dev)))
