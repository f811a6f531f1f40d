"""
network -- the policy/value net of /root/reference/network.py:15-198, state_dict-compatible.

Same tensors, same keys (conv_input, bn_input, residual_tower.{i}.{conv1,bn1,conv2,bn2[,seblock.excitation.{0,2}]},
policy_conv, policy_bn, policy_fc, value_conv, value_bn, value_fc1, value_fc2), so reference checkpoints
load with load_state_dict (main.py:47-49, uci.py:37).  The evaluate stage is the only MFMA work on the
path and runs under PyTorch-ROCm (MIOpen / hipBLASLt); `for_inference()` gives the rollout engine a
BN-folded, channels-last copy (eval-mode BatchNorm is an affine map, so folding changes results only by
float rounding).
"""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

import config


class SEBlock(nn.Module):
    """Squeeze-and-excitation gate (network.py:15-45): global average -> FC -> ReLU -> FC -> sigmoid."""

    def __init__(self, num_channels: int, reduction_ratio: int = 16):
        super().__init__()
        self.channels, self.reduction_ratio = num_channels, reduction_ratio
        hidden = num_channels // reduction_ratio
        self.squeeze = nn.AdaptiveAvgPool2d(1)
        self.excitation = nn.Sequential(nn.Linear(num_channels, hidden, bias=False), nn.ReLU(inplace=True),
                                        nn.Linear(hidden, num_channels, bias=False), nn.Sigmoid())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gate = self.excitation(self.squeeze(x).flatten(1))
        return x * gate[:, :, None, None]


class ResidualBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN + skip, ReLU (network.py:48-80)."""

    def __init__(self, num_filters: int):
        super().__init__()
        self.conv1 = nn.Conv2d(num_filters, num_filters, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(num_filters)
        self.conv2 = nn.Conv2d(num_filters, num_filters, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(num_filters)

    def _body(self, x):
        return self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.relu(self._body(x) + x)


class SEResidualBlock(ResidualBlock):
    """Residual block whose body is gated by an SEBlock before the skip (network.py:83-118)."""

    def __init__(self, num_filters: int):
        super().__init__(num_filters)
        self.seblock = SEBlock(num_filters, config.SE_REDUCTION_RATIO)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.relu(self.seblock(self._body(x)) + x)


class PolicyValueNet(nn.Module):
    """network.py:121-198.  Sizes come from config at construction (network.py:130-143)."""

    def __init__(self):
        super().__init__()
        f = config.CONV_FILTERS
        self.conv_input = nn.Conv2d(config.INPUT_CHANNELS, f, 3, padding=1, bias=False)
        self.bn_input = nn.BatchNorm2d(f)
        blocks = [ResidualBlock(f) for _ in range(config.RESIDUAL_BLOCKS)]
        blocks += [SEResidualBlock(f) for _ in range(config.SE_RESIDUAL_BLOCKS)]
        self.residual_tower = nn.Sequential(*blocks)
        cells = config.BOARD_SIZE * config.BOARD_SIZE
        self.policy_conv = nn.Conv2d(f, 2, 1, bias=False)
        self.policy_bn = nn.BatchNorm2d(2)
        self.policy_fc = nn.Linear(2 * cells, config.NUM_ACTIONS)
        self.value_conv = nn.Conv2d(f, 32, 1, bias=False)
        self.value_bn = nn.BatchNorm2d(32)
        self.value_fc1 = nn.Linear(32 * cells, 256)
        self.value_fc2 = nn.Linear(256, 1)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = self.residual_tower(F.relu(self.bn_input(self.conv_input(x))))
        p = F.relu(self.policy_bn(self.policy_conv(x))).flatten(1)
        v = F.relu(self.value_bn(self.value_conv(x))).flatten(1)
        return self.policy_fc(p), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))

    @torch.no_grad()
    def for_inference(self, dtype: torch.dtype = torch.float32, channels_last: bool = True) -> nn.Module:
        """Eval-mode copy with every BatchNorm folded into the convolution in front of it."""
        import copy

        net = copy.deepcopy(self).eval()

        def fold(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, padding=conv.padding, bias=True)
            out.weight.copy_(conv.weight * scale[:, None, None, None])
            out.bias.copy_(bn.bias - bn.running_mean * scale)
            return out.to(conv.weight.device)

        net.conv_input, net.bn_input = fold(net.conv_input, net.bn_input), nn.Identity()
        for blk in net.residual_tower:
            blk.conv1, blk.bn1 = fold(blk.conv1, blk.bn1), nn.Identity()
            blk.conv2, blk.bn2 = fold(blk.conv2, blk.bn2), nn.Identity()
        net.policy_conv, net.policy_bn = fold(net.policy_conv, net.policy_bn), nn.Identity()
        net.value_conv, net.value_bn = fold(net.value_conv, net.value_bn), nn.Identity()
        net = net.to(dtype)
        if channels_last:
            net = net.to(memory_format=torch.channels_last)
        for p in net.parameters():
            p.requires_grad_(False)
        return net
