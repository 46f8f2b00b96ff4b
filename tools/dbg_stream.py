"""One convolution through ops.conv2d against torch (debugging the streaming kernel): python tools/dbg_stream.py B Cin H W Cout k stride pad"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack
B, Cin, H, W, Cout, k, stride, pad = [int(v) for v in sys.argv[1:9]]
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W)
w = torch.randn(Cout, Cin, k, k) * 0.05
b = torch.randn(Cout)
want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad).float()
pc = pack.pack_conv(w, b, stride=stride, padding=pad).to("cuda")
xd = ops.to_nhwc(x.cuda())
got = ops.to_nchw(ops.conv2d(xd, pc))
torch.cuda.synchronize()
print("ok", sys.argv[1:], os.environ.get("ZS_STREAM_FORCE"), "max err %.3e of %.3e" % (float((got.cpu() - want).abs().max()), float(want.abs().max())))
g, w_ = got.cpu()[0], want[0]                      # [Cout][H][W]
d = (g - w_).abs()
bad = torch.isnan(g) | (d > 1e-3 * float(w_.abs().max()))
print("bad fraction %.3f; nan %d" % (float(bad.float().mean()), int(torch.isnan(g).sum())))
print("bad per channel block of 8:", [int(bad[c:c + 8].sum()) for c in range(0, min(Cout, 64), 8)])
pix = bad.reshape(Cout, -1).any(0)
print("bad pixels:", [i for i in range(pix.numel()) if pix[i]][:40], "of", pix.numel())
bm = bad.reshape(Cout, -1)
import math
print("bad count per (32-channel tile, 32-pixel tile):")
for ct in range(0, Cout, 32):
    print("  ch %3d:" % ct, [int(bm[ct:ct + 32, pt:pt + 32].sum()) for pt in range(0, bm.shape[1], 32)])
