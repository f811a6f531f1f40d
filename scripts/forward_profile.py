#!/usr/bin/env python3
"""scripts/forward_profile.py -- one evaluate-stage forward (10x128 net, 256 boards) replayed under a hipGraph; run under
rocprofv3 --kernel-trace --stats to see which kernels make up a forward.  usage: forward_profile.py [conv] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet

conv = sys.argv[1] if len(sys.argv) > 1 else "tower_wg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = (15, 5, 256) if conv == "tower_f16" else (8, 2, 128)
torch.manual_seed(0)
net = FusedPolicyValueNet(network.PolicyValueNet().cuda().eval(), conv=conv).cuda()
x = torch.randn(B, 120, 8, 8, device="cuda")
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(x)
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
print("done", out[0].shape)
