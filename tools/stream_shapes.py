"""Time one pointwise layer shape of the batch-1 encoder through ops.conv2d, inside a captured graph of `reps` back-to-back
launches that alternate between two weight sets and ping-pong their activations (cold-ish L2, like the real forward):

    python tools/stream_shapes.py M K N [stats] [reps]

Run it under different environments to compare the kernels / plans:
    ZS_CONV_STREAM=0                      small-tile kernel (with its two-launch K split when the caller allows it)
    ZS_STREAM_FORCE=mi,nj,z               streaming kernel with a forced tile shape and K split
    ZS_STREAM_SPLIT=1                     streaming kernel with the in-launch K split allowed (off by default)
    ZS_STREAM_MIN_TILES=1                 ... and no lower bound on the number of workgroups
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack
from zeroshape_amd import _lib

M, K, N = [int(v) for v in sys.argv[1:4]]
stats = len(sys.argv) > 4 and sys.argv[4] == "stats"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
torch.manual_seed(0)
dev = torch.device("cuda:0")
layers = []
for i in range(8):                       # eight weight sets: 8 x K x N x 4 B of weights keep the L2s from holding any one
    w = torch.randn(N, K, 1, 1) / K ** 0.5
    layers.append(pack.pack_conv(w, torch.randn(N)).to(dev))
xs = [torch.randn(1, 1, M, K, device=dev) for _ in range(8)]
res = torch.randn(1, 1, M, N, device=dev)


def run():
    out = None
    for r in range(reps):
        pc, x = layers[r % 8], xs[r % 8]
        if stats:
            out, st = ops.conv2d(x, pc, res1=res, stats_out="row")
        else:
            out = ops.conv2d(x, pc, res1=res)
    return out


want = run().clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        got = run()
torch.cuda.synchronize()
best = 1e9
for trial in range(5):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    best = min(best, t0.elapsed_time(t1) * 1e3 / (5 * reps))
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("ZS_CONV_STREAM", "ZS_STREAM_FORCE", "ZS_STREAM_SPLIT", "ZS_STREAM_MIN_TILES") if k in os.environ)
print("M %4d K %4d N %4d %-5s  %-40s %7.2f us per launch   max |diff| vs first run %.1e   calls %d" %
      (M, K, N, "stats" if stats else "", tag or "default", best, float((got - want).abs().max()), _lib.CALLS[0]), flush=True)
