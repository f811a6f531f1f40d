#!/usr/bin/env python3
"""scripts/forward_profile_b1.py -- the batch-1 forward of the 20x256 net (uci.py) replayed under a hipGraph; run under rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
torch.manual_seed(0)
net = FusedPolicyValueNet(network.PolicyValueNet().cuda().eval(), conv="mfma_small").cuda()
x = torch.randn(1, 120, 8, 8, device="cuda")
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(x)
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
print("done")
