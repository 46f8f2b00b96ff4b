"""GroupNorm inference kernel per layer shape of the batch-28 encoder (ZS_GN_ONE_LAUNCH=1: the one-launch kernels)."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zeroshape_amd.nn import ops
for shape in ((28, 56, 56, 256), (28, 56, 56, 64), (28, 28, 28, 512), (28, 28, 28, 128), (28, 14, 14, 1024), (28, 14, 14, 256), (28, 112, 112, 64)):
    B, H, W, C = shape
    x = torch.randn(*shape, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    # the producer would have just written x: emulate by touching it
    def run():
        return ops.group_norm(x, g, b, 32, 1e-5, True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        run(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[2]
    mb = x.numel() * 4 / 1e6
    print(shape, "%.1f MB  %.1f us  %.2f TB/s (rd+wr)" % (mb, ms * 1e3, 2 * mb / (ms * 1e-3) / 1e6), flush=True)
