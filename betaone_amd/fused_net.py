"""
betaone_amd/fused_net.py -- the evaluate stage with hand-written fused epilogues.

Same function as PolicyValueNet.forward (/root/reference/network.py:167-198) in eval mode: BatchNorm is folded into
the convolutions, the 3x3 convolutions stay in MIOpen (fp32 Winograd asm kernel under PyTorch-ROCm, the only MFMA/
matrix work on the path), and every conv is followed by ONE gfx950 kernel from csrc/bo_nn_fused.h instead of the
3-8 separate elementwise launches PyTorch issues (bias, ReLU, residual add, SE pooling / FCs / sigmoid / scale).
NCHW float32 only; other dtypes/layouts use PolicyValueNet.for_inference().
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine as E


class FusedPolicyValueNet(nn.Module):
    def __init__(self, net, lib=None):
        """net: a PolicyValueNet (any device); weights are copied, BN folded."""
        super().__init__()
        self.lib = lib if lib is not None else E.load_hip_library()
        f = net.for_inference(dtype=torch.float32, channels_last=False)
        dev = next(f.parameters()).device
        if dev.type != "cuda":
            raise E.EngineError("FusedPolicyValueNet needs the net on an MI355X (cuda device)")

        def cw(conv):
            return nn.Parameter(conv.weight.detach().contiguous(), requires_grad=False), \
                nn.Parameter(conv.bias.detach().contiguous(), requires_grad=False)

        self.w_in, self.b_in = cw(f.conv_input)
        self.blocks = []
        for i, blk in enumerate(f.residual_tower):
            w1, b1 = cw(blk.conv1)
            w2, b2 = cw(blk.conv2)
            se = None
            if hasattr(blk, "seblock"):
                se = (nn.Parameter(blk.seblock.excitation[0].weight.detach().contiguous(), requires_grad=False),
                      nn.Parameter(blk.seblock.excitation[2].weight.detach().contiguous(), requires_grad=False))
            for k, p in (("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2)):
                self.register_parameter(f"blk{i}_{k}", p)
            if se:
                self.register_parameter(f"blk{i}_se1", se[0])
                self.register_parameter(f"blk{i}_se2", se[1])
            self.blocks.append((w1, b1, w2, b2, se))
        # the two 1x1 head convolutions share their input: one conv with 2 + 32 output channels
        self.w_head = nn.Parameter(torch.cat([f.policy_conv.weight, f.value_conv.weight], 0).detach().contiguous(), requires_grad=False)
        self.b_head = nn.Parameter(torch.cat([f.policy_conv.bias, f.value_conv.bias], 0).detach().contiguous(), requires_grad=False)
        self.n_policy_ch = f.policy_conv.out_channels
        self.policy_fc, self.value_fc1, self.value_fc2 = f.policy_fc, f.value_fc1, f.value_fc2
        self.layout = "nchw+fused"

    def _epi(self, x, bias, res=None):
        B, C = x.shape[0], x.shape[1]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = self.lib.bo_nn_bias_act(x.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, B, C, stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return x

    def _se(self, x, bias, se, res):
        B, C = x.shape[0], x.shape[1]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = self.lib.bo_nn_se_residual(x.data_ptr(), bias.data_ptr(), se[0].data_ptr(), se[1].data_ptr(), res.data_ptr(), B, C,
                                        se[0].shape[0], stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return x

    @torch.no_grad()
    def forward(self, x):
        x = self._epi(F.conv2d(x, self.w_in, None, padding=1), self.b_in)
        for w1, b1, w2, b2, se in self.blocks:
            y = self._epi(F.conv2d(x, w1, None, padding=1), b1)
            y = F.conv2d(y, w2, None, padding=1)
            x = self._se(y, b2, se, x) if se is not None else self._epi(y, b2, x)
        h = self._epi(F.conv2d(x, self.w_head, None), self.b_head)
        p = h[:, :self.n_policy_ch].flatten(1)
        v = h[:, self.n_policy_ch:].flatten(1)
        return self.policy_fc(p), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
