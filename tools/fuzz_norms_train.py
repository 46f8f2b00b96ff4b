#!/usr/bin/env python3
"""Random-shape sweep of the training normalisation / attention kernels against torch's CPU autograd, through the
repository's own tests as checkers: BatchNorm (one-launch strip kernel up to 1,024 rows, chunked above: shapes on
both sides of the switch), GroupNorm backward (pixel slices), LayerNorm backward (wide partial reduce), point
attention (MFMA forward / backward).    python tools/fuzz_norms_train.py [N] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train_encoder import test_batch_norm_train as bn, test_group_norm_backward as gn     # noqa: E402
from tests.test_gpu_train_ops import test_layer_norm_backward as ln, test_point_attention as pa          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def run(name, fn, *args):
    global bad
    try:
        fn(*args)
    except AssertionError as e:
        bad += 1
        print("FAIL", name, args, str(e)[:160])


for i in range(n):
    B, H = rnd.choice([1, 2, 3, 4, 5]), rnd.choice([1, 2, 7, 9, 14, 16, 18, 23, 28, 33])
    B = max(B, 2) if H == 1 else B               # torch refuses batch statistics over a single row
    run("batch_norm", bn, B, H, 4 * rnd.randint(1, 260), rnd.random() < 0.5, rnd.random() < 0.5)
    run("group_norm", gn, rnd.choice([1, 2, 3, 4]), rnd.choice([3, 7, 8, 14, 20, 28, 31, 56]), 32 * rnd.choice([1, 2, 4, 8, 16, 32]),
        rnd.random() < 0.5, rnd.random() < 0.5)
    run("layer_norm", ln, rnd.randint(1, 5000), 64 * rnd.randint(1, 16))
    run("point_attention", pa, rnd.choice([1, 2, 3]), rnd.randint(1, 2500), rnd.randint(1, 256))
print("%d rounds of 4 kernels, %d failures" % (n, bad))
sys.exit(1 if bad else 0)
