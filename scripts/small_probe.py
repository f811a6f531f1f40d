#!/usr/bin/env python3
"""scripts/small_probe.py -- batch-1 forward of the 20x256 net (uci.py's case) under a hipGraph: MIOpen vs bo_k_conv3x3_small."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet, conv3x3_small, pack_conv_weight_small
from betaone_amd import engine as E
import torch.nn.functional as F

def bench(net, x, n=50):
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:
            net(x); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            net(x)
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
torch.manual_seed(0)
base = network.PolicyValueNet().cuda().eval()
lib = E.load_hip_library()
for B in (1, 4):
    x = torch.randn(B, 120, 8, 8, device="cuda")
    for cl in (False, True):
        net = base.for_inference(dtype=torch.float32, channels_last=cl)
        xx = x.contiguous(memory_format=torch.channels_last) if cl else x
        print(f"B={B} MIOpen channels_last={cl}: {bench(net, xx):.3f} ms", flush=True)
    for conv in ("miopen", "mfma_small"):
        net = FusedPolicyValueNet(base, conv=conv).cuda()
        print(f"B={B} fused conv={conv}: {bench(net, x):.3f} ms", flush=True)
    xc = torch.randn(B, 256, 8, 8, device="cuda"); w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
    bias = torch.zeros(256, device="cuda"); wp = pack_conv_weight_small(w)
    class One(torch.nn.Module):
        def forward(self, x):
            for _ in range(10): x = conv3x3_small(lib, x, wp, bias, 256, 256, 1)
            return x
    class Lib(torch.nn.Module):
        def forward(self, x):
            for _ in range(10): x = F.conv2d(x, w, None, padding=1)
            return x
    print(f"B={B} 10 x conv 256->256: small {bench(One(), xc)*100:.1f} us/layer   MIOpen {bench(Lib(), xc)*100:.1f} us/layer", flush=True)
