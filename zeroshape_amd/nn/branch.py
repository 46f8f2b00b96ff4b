"""Independent branches of the encoder's launch graph on side streams.

At batch 1 the encoder is a chain of ~300 launches that each leave most of the 256 CUs idle (a 14x14 map or a 197-token
matrix fills 30-250 workgroups and waits on its own dependent loads), so two launches that do not depend on each other
cost little more together than either alone.  The network has such pairs: the skip convolutions of DPT's decoder (they
need only the backbone taps), the intrinsics head beside the fusion blocks, the reassemble branch of tap 3 beside ViT
blocks 9-11, the ResNet down-sampling shortcuts beside the residual branch, depth_feat_proj beside layer4.

    br = Branch(x)                 # side stream waits for everything queued on the current stream
    with br:                       # launches inside go to the side stream
        y = ops.conv2d(x, pc)
    ...                            # the current stream continues meanwhile
    y = br.join(y)                 # the current stream waits for the side stream

Under stream capture (nn/capture.py) the waits become graph edges, so the captured hipGraph holds the branches as
parallel paths.  Memory: the caching allocator orders reuse of a block only against the stream that allocated it, so a
tensor one stream allocates and another reads is kept alive until the join (the Branch holds the references it is given);
`join` orders the other direction.

MEASURED (round 4, tools/enc_b1.py, batch 1, replayed hipGraph): every fork is a LOSS - one stream 3.515 ms; intrinsics head
forked 3.70; skip convolutions 3.81; depth_feat_proj 3.77; shortcuts 3.81; all four 3.82 ms (outputs bit-identical).  A
hipGraph with parallel paths pays more per cross-stream edge than two ~10 us launches gain by overlapping, so the forks are
OFF by default (ZS_BRANCH_KINDS=0); ZS_BRANCH_KINDS=<mask> (1 shortcuts, 2 skips, 4 intrinsics head, 8 depth_feat_proj)
re-enables them for measurements."""
import os

import torch

ENABLED = os.environ.get("ZS_BRANCHES", "1") != "0"
# which forks are taken (bit mask, A/B measurements): 1 = ResNet projection shortcuts, 2 = DPT skip convolutions,
# 4 = intrinsics head, 8 = depth_feat_proj
KINDS = int(os.environ.get("ZS_BRANCH_KINDS", "0"))
SHORTCUT, SKIP, INTR, PROJ = 1, 2, 4, 8
_POOL = {}
_BUSY = set()


def _side_stream(device, slot):
    key = (str(device), slot)
    if key not in _POOL:
        _POOL[key] = torch.cuda.Stream(device=device)
    return _POOL[key]


def warm(device, slots=6):
    """Create the side streams of `device` (stream creation is not allowed inside a capture)."""
    if ENABLED:
        for d in range(slots):
            _side_stream(device, d)


class Branch:
    """One side branch.  Branches open at the same time run on different streams of the pool; a stream returns to the
    pool at the join."""

    def __init__(self, *inputs, kind=0xFFFF):
        self.keep = [t for t in inputs if torch.is_tensor(t)]
        self.on = ENABLED and (KINDS & kind) != 0 and len(self.keep) > 0 and self.keep[0].is_cuda
        if self.on:
            dev = self.keep[0].device
            self.slot = next(i for i in range(64) if (str(dev), i) not in _BUSY)
            _BUSY.add((str(dev), self.slot))
            self.side = _side_stream(dev, self.slot)
            self.side.wait_stream(torch.cuda.current_stream(dev))
            self.ctx = None

    def __enter__(self):
        if self.on:
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def hold(self, *tensors):
        """Keep tensors the branch reads (allocated by another stream) alive until the join."""
        self.keep.extend(t for t in tensors if torch.is_tensor(t))

    def join(self, *outputs):
        """The current stream waits for the branch; returns the outputs (now safe to use on the current stream)."""
        if self.on:
            dev = self.keep[0].device
            torch.cuda.current_stream(dev).wait_stream(self.side)
            # the branch's outputs were allocated by the side stream: every later use of that stream starts with a wait
            # on the then-current stream (__init__), which orders their reuse behind their readers
            _BUSY.discard((str(dev), self.slot))
            self.on = False
        self.keep = []
        if len(outputs) == 1:
            return outputs[0]
        return outputs
