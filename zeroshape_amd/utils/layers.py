"""Parameter containers with the reference's utils/layers.py names.  Their arithmetic runs in
zeroshape_amd/nn (HIP); calling them directly raises."""
import torch.nn as nn


class Bottleneck_Conv(nn.Module):
    """utils/layers.py:76-100: conv-BN-ReLU-conv-BN, residual, ReLU (keys linear1, bn1, linear2,
    bn2).  Executed by nn.blocks.run_bottleneck_conv."""

    def __init__(self, n_channels, kernel_size=1):
        super().__init__()
        self.kernel_size = kernel_size
        self.linear1 = nn.Conv2d(n_channels, n_channels, kernel_size=kernel_size, padding=kernel_size // 2, bias=False)
        self.bn1 = nn.BatchNorm2d(n_channels)
        self.linear2 = nn.Conv2d(n_channels, n_channels, kernel_size=kernel_size, padding=kernel_size // 2, bias=False)
        self.bn2 = nn.BatchNorm2d(n_channels)

    def forward(self, x):
        raise RuntimeError("Bottleneck_Conv is a parameter container; its owner runs it on the HIP library")
