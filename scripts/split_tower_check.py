#!/usr/bin/env python3
"""scripts/split_tower_check.py -- the split-precision tower (csrc/bo_tower_s.h, conv='tower_split') against the float32 towers:
error table against a float64 evaluation of the same net, and event-timed forwards.
usage: split_tower_check.py [batch ...]        (prints markdown; default batches 256 1024 4096)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from betaone_amd import dropin

dropin.install()
import config
import network
from betaone_amd.fused_net import FusedPolicyValueNet


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


def main():
    batches = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g1_net.npz"))
    base = torch.from_numpy(z["inputs"]).cuda()
    print("| net | conv | boards | max abs err logits vs f64 | max abs err value vs f64 | tower us (median of 30) |")
    print("|---|---|---|---|---|---|")
    for size, name in (((8, 2, 128), "8+2x128"), ((15, 5, 256), "15+5x256")):
        config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = size
        torch.manual_seed(1)
        net = network.PolicyValueNet().eval().cuda()
        with torch.no_grad():  # BatchNorm statistics away from the identity, so that folding matters
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.uniform_(-0.2, 0.2)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.uniform_(-0.2, 0.2)
        net64 = network.PolicyValueNet().eval().cuda().double()
        net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in net.state_dict().items()})
        convs = ["tower_split"] + (["tower_wg"] if size[2] == 128 else ["mfma"])
        for conv in convs:
            fused = FusedPolicyValueNet(net, conv=conv).cuda()
            for B in batches:
                x = (base.repeat(B // 3 + 1, 1, 1, 1)[:B] * torch.linspace(0.25, 1.0, B, device="cuda")[:, None, None, None]).contiguous()
                with torch.no_grad():
                    l64, v64 = net64(x.double())
                    l, v = fused(x)
                    el = (l.double() - l64).abs().max().item()
                    evv = (v.double() - v64).abs().max().item()
                    if conv in ("tower_split", "tower_wg"):
                        us = timed(lambda: fused._tower_forward(x, heads=True))
                    else:
                        us = timed(lambda: fused._tower_mfma(x))
                print(f"| {name} | {conv} | {B} | {el:.3e} | {evv:.3e} | {us:.1f} |", flush=True)
            del fused
    # fp16 subnormal operands: a net whose first-layer weights are tiny must still come out right (the lo halves of its activations
    # are subnormal fp16 numbers; a matrix pipe that flushed them would lose ~2^-14 absolute per activation)
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = (2, 1, 128)
    torch.manual_seed(2)
    net = network.PolicyValueNet().eval().cuda()
    with torch.no_grad():
        net.conv_input.weight.mul_(2.0 ** -10)
    net64 = network.PolicyValueNet().eval().cuda().double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in net.state_dict().items()})
    fused = FusedPolicyValueNet(net, conv="tower_split").cuda()
    x = base.repeat(11, 1, 1, 1).contiguous()
    with torch.no_grad():
        l64, v64 = net64(x.double())
        l, v = fused(x)
        l32, v32 = net(x)
    print()
    print(f"small activations (input conv weights x 2^-10): tower_split max abs err logits {(l.double() - l64).abs().max().item():.3e} "
          f"(float32 torch net: {(l32.double() - l64).abs().max().item():.3e}), logits magnitude {l64.abs().max().item():.3e}")


if __name__ == "__main__":
    main()
