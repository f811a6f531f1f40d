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
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:  # idle clocks after host-side weight packing
            net(x); torch.cuda.synchronize()
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
for bm in (() if '--fused-only' in sys.argv else (False, True)):
    torch.backends.cudnn.benchmark = bm
    for cl in (True, False):
        for dt in (torch.float32, torch.float16):
            net = base.for_inference(dtype=dt, channels_last=cl)
            for B in (256, 512, 1024, 2048):
                x = torch.randn(B, 120, 8, 8, device="cuda", dtype=dt)
                if cl: x = x.contiguous(memory_format=torch.channels_last)
                ms = bench(net, x)
                print(f"benchmark={bm} channels_last={cl} dtype={str(dt)[6:]} B={B}: {ms:.3f} ms  {B/ms*1e3:.0f} pos/s  {gflop*B/ms:.1f} TFLOP/s", flush=True)

# hand-written variants (csrc/bo_nn_fused.h, csrc/bo_conv.h), NCHW fp32
from betaone_amd.fused_net import FusedPolicyValueNet, conv3x3_mfma, pack_conv_weight
from betaone_amd import engine as E
import torch.nn.functional as F
lib = E.load_hip_library()
for conv in ("miopen", "mfma", "tower", "tower_wg"):
    net = FusedPolicyValueNet(base, conv=conv).cuda()
    for B in (256, 512, 1024, 2048):
        x = torch.randn(B, 120, 8, 8, device="cuda")
        ms = bench(net, x)
        print(f"fused conv={conv} B={B}: {ms:.3f} ms  {B/ms*1e3:.0f} pos/s  {gflop*B/ms:.1f} TFLOP/s", flush=True)
for kind in ("tower", "tower_wg"):
    tw = FusedPolicyValueNet(base, conv=kind).cuda()
    class TowerOnly(torch.nn.Module):
        def forward(self, x): return tw._tower_forward(x)
    for B in (128, 256, 512):
        x = torch.randn(B, 120, 8, 8, device="cuda")
        ms = bench(TowerOnly(), x, 100)
        print(f"{kind} kernel alone B={B}: {ms*1e3:.1f} us = {ms*1e3/21:.1f} us/layer", flush=True)
for B in (256, 1024):
    x = torch.randn(B, 128, 8, 8, device="cuda"); w = torch.randn(128, 128, 3, 3, device="cuda") * 0.03
    bias = torch.zeros(128, device="cuda"); wp = pack_conv_weight(w); y = torch.empty_like(x)
    class One(torch.nn.Module):
        def forward(self, x): return conv3x3_mfma(lib, x, wp, bias, 128, 1, out=y)
    class Lib(torch.nn.Module):
        def forward(self, x): return F.conv2d(x, w, None, padding=1)
    for nm, m in (("mfma", One()), ("miopen", Lib())):
        ms = bench(m, x, 100)
        print(f"single conv 128->128 {nm} B={B}: {ms*1e3:.1f} us  {B*128*128*9*64*2/ms/1e9:.1f} TFLOP/s", flush=True)

# fp16 tower (csrc/bo_tower_h.h) vs MIOpen fp16 on the 20x256 net of BASELINE configs[4]
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
big = network.PolicyValueNet().cuda().eval()
f16 = FusedPolicyValueNet(big, conv="tower_f16").cuda()
for cl in (True, False):
    hn = big.for_inference(dtype=torch.float16, channels_last=cl)
    for B in (256, 512, 1024):
        x = torch.randn(B, 120, 8, 8, device="cuda", dtype=torch.float16)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        ms = bench(hn, x)
        print(f"20x256 fp16 MIOpen channels_last={cl} B={B}: {ms:.3f} ms  {B/ms*1e3:.0f} pos/s", flush=True)
for B in (256, 512, 1024):
    x = torch.randn(B, 120, 8, 8, device="cuda")
    ms = bench(f16, x)
    print(f"20x256 fp16 tower B={B}: {ms:.3f} ms  {B/ms*1e3:.0f} pos/s  {41*2*9*256*256*64*B/ms/1e9:.0f} TFLOP/s", flush=True)
