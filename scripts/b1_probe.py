#!/usr/bin/env python3
"""scripts/b1_probe.py -- LAB: the batch-1 evaluation of the 15+5x256 net (uci.py, BASELINE.json configs[3]) through conv='tower_b1'
(one launch for the tower, csrc/bo_tower_b1.h) and conv='mfma_small' (one launch per layer): agreement and time per evaluation,
eager and replayed from a graph.  usage: b1_probe.py [plain blocks] [se blocks] [filters] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet

a = [int(v) for v in sys.argv[1:]] + [15, 5, 256, 1][len(sys.argv) - 1:]
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = a[0], a[1], a[2]
B = a[3]
torch.manual_seed(0)
plain = network.PolicyValueNet().cuda().eval()
nets = {"tower_b1": FusedPolicyValueNet(plain, conv="tower_b1", f32_pipe=os.environ.get("B1_F32", "0") == "1").cuda(),
        "mfma_small": FusedPolicyValueNet(plain, conv="mfma_small").cuda()}
print("tower_b1 precision:", nets["tower_b1"].b1_precision)
x = torch.rand(B, 120, 8, 8, device="cuda")
with torch.no_grad():
    ref = nets["mfma_small"]._tower_small(x)
    got = nets["tower_b1"]._tower_b1(x)
    torch.cuda.synchronize()
    nets["tower_b1"].check_b1()
    print("tower: max |b1 - per layer| =", (got - ref).abs().max().item(), " max |ref| =", ref.abs().max().item())
    l0, v0 = plain(x)
    for k, net in nets.items():
        l, v = net(x)
        print(k, "vs torch: logits", (l - l0).abs().max().item(), "value", (v - v0).abs().max().item())
    for k, net in nets.items():
        for what in ("tower", "forward"):
            fn = (lambda: net._tower_b1(x)) if (what == "tower" and k == "tower_b1") else (lambda: net._tower_small(x)) if what == "tower" else (lambda: net(x))
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 200
            e0.record()
            for _ in range(n):
                g.replay()
            e1.record(); e1.synchronize()
            print(f"{k:11s} {what:8s} graph replay: {e0.elapsed_time(e1) * 1e3 / n:8.1f} us per evaluation")
    nets["tower_b1"].check_b1()
    # per-wave phase clocks of a layer (eager launches: the graphs above hold the unprofiled argument)
    import ctypes as C
    import numpy as np
    net = nets["tower_b1"]
    net.lib.bo_nn_b1_profile(net._b1, 1, None, 0)
    reps = 50
    for _ in range(reps):
        net._tower_b1(x)
    torch.cuda.synchronize()
    tiles = (a[2] // 16) * 4
    out = np.zeros((B * tiles * 4, 8), dtype=np.uint64)
    net.lib.bo_nn_b1_profile(net._b1, 0, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.shape[0])
    p = out.reshape(B * tiles, 4, 8).astype(np.float64)
    layers = p[:, :, 5].max()
    names = ["wait", "stage", "mfma", "reduce", "epilogue+signal"]
    for w in range(4):
        row = "  ".join(f"{names[k]} {p[:, w, k].mean() / layers:8.0f}" for k in range(5))
        print(f"wave {w}: clocks per layer (mean over {B * tiles} workgroups, {reps} launches): {row}   sum {p[:, w, :5].sum(axis=1).mean() / layers:8.0f}")
print("done")
