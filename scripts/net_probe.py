#!/usr/bin/env python3
"""scripts/net_probe.py -- forward time of the policy/value net under PyTorch-ROCm for a few settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network

def bench(net, x, n=30):
    with torch.no_grad():
        for _ in range(5): net(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            net(x)
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
torch.manual_seed(0)
base = network.PolicyValueNet().cuda().eval()
gflop = 0.398
for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    for cl in (True, False):
        for dt in (torch.float32, torch.float16):
            net = base.for_inference(dtype=dt, channels_last=cl)
            for B in (256, 512, 1024, 2048):
                x = torch.randn(B, 120, 8, 8, device="cuda", dtype=dt)
                if cl: x = x.contiguous(memory_format=torch.channels_last)
                ms = bench(net, x)
                print(f"benchmark={bm} channels_last={cl} dtype={str(dt)[6:]} B={B}: {ms:.3f} ms  {B/ms*1e3:.0f} pos/s  {gflop*B/ms:.1f} TFLOP/s", flush=True)
