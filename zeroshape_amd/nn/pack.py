"""Host-side parameter packing for csrc/nn_conv.hip (done once per checkpoint).

A convolution weight [Cout][Cin][kh][kw] (torch layout) becomes the GEMM operand
[K16/4][CoutPad][4] with k = (ky*kw + kx)*Cin + cin, zero padded to K16 = ceil16(K) and
CoutPad = ceil128(Cout) (include/zeroshape_hip.h, zs_conv2d_nhwc).  BatchNorm in eval mode and
biases fold into the per-channel (scale, shift) of the epilogue."""
import math

import torch

BK, BN = 16, 128


class PackedConv:
    """Geometry + packed parameters of one convolution / linear layer."""

    def __init__(self, w, scale, shift, cin, cout, kh, kw, stride, padding):
        self.w, self.scale, self.shift = w, scale, shift
        self.w16 = None                     # split-fp16 halves of w (ops.conv2d, built on first f16x3 use)
        self.cin, self.cout, self.kh, self.kw, self.stride = cin, cout, kh, kw, stride
        self.padding = padding              # int (torch symmetric zero padding) or "same" (timm / TF)

    def out_size(self, n, k):
        """(output length, leading pad) along one axis of input length n and kernel k."""
        if self.padding == "same":
            out = -(-n // self.stride)
            total = max((out - 1) * self.stride + k - n, 0)
            return out, total // 2
        return (n + 2 * self.padding - k) // self.stride + 1, self.padding

    def to(self, device):
        self.w = self.w.to(device)
        self.w16 = None
        self.scale = None if self.scale is None else self.scale.to(device)
        self.shift = None if self.shift is None else self.shift.to(device)
        return self


def pack_weight(weight, cin_pad=None):
    """[Cout][Cin][kh][kw] -> flat fp32 [K16/4][CoutPad][4]."""
    w = weight.detach().to(torch.float32).cpu()
    cout, cin, kh, kw = w.shape
    if cin_pad is not None and cin_pad > cin:
        w = torch.cat([w, torch.zeros(cout, cin_pad - cin, kh, kw)], 1)
        cin = cin_pad
    assert cin % 4 == 0, "Cin must be a multiple of 4 (pad the input channels)"
    K = kh * kw * cin
    K16 = -(-K // BK) * BK
    coutp = -(-cout // BN) * BN
    wk = w.permute(2, 3, 1, 0).reshape(K, cout)            # [k][cout], k = tap*Cin + cin
    full = torch.zeros(K16, coutp)
    full[:K, :cout] = wk
    return full.view(K16 // 4, 4, coutp).permute(0, 2, 1).contiguous().view(-1), cin


def fold_bn(bn, bias=None):
    """Eval-mode BatchNorm (+ an optional preceding bias) as per-channel (scale, shift)."""
    g, b, mean, var, eps = [bn[k] for k in ("weight", "bias", "running_mean", "running_var", "eps")]
    scale = g.double() / torch.sqrt(var.double() + eps)
    shift = b.double() - mean.double() * scale
    if bias is not None:
        shift = shift + bias.double() * scale
    return scale.float(), shift.float()


def pack_conv(weight, bias=None, bn=None, stride=1, padding=0, cin_pad=None):
    """weight [Cout][Cin][kh][kw] (or [Cout][Cin] for a Linear), optional bias, optional eval-mode
    BatchNorm dict(weight, bias, running_mean, running_var, eps)."""
    if weight.dim() == 2:
        weight = weight[:, :, None, None]
    cout, _, kh, kw = weight.shape
    w, cin = pack_weight(weight, cin_pad)
    if bn is not None:
        scale, shift = fold_bn(bn, bias)
    else:
        scale, shift = None, (None if bias is None else bias.detach().float().cpu().clone())
    return PackedConv(w, scale, shift, cin, cout, kh, kw, stride, padding)


def standardize_weight(weight, eps):
    """timm StdConv2d / StdConv2dSame (timm==0.6.12, layers/std_conv.py): per output channel
    (w - mean) / sqrt(biased var + eps), as F.batch_norm(training=True) computes it."""
    w = weight.detach().double()
    flat = w.reshape(w.shape[0], -1)
    mean = flat.mean(1, keepdim=True)
    var = flat.var(1, unbiased=False, keepdim=True)
    return ((flat - mean) / torch.sqrt(var + eps)).reshape(w.shape).float()
