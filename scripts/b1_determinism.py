#!/usr/bin/env python3
"""scripts/b1_determinism.py -- LAB: the one-launch tower (conv='tower_b1') must give a board the same bits whatever the batch it
rides in and on every repetition (per-game results of self-play may not depend on the slot count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet

for size in ((3, 1, 64), (8, 2, 128), (15, 5, 256)):
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = size
    torch.manual_seed(0)
    plain = network.PolicyValueNet().cuda().eval()
    for f32 in (True, False):
        net = FusedPolicyValueNet(plain, conv="tower_b1", f32_pipe=f32).cuda()
        bmax = net._b1_max
        x = torch.rand(bmax, 120, 8, 8, device="cuda")
        with torch.no_grad():
            ref = net._tower_b1(x).clone()
            bad_rep = bad_sub = bad_fwd = 0
            l0, v0 = net(x)
            for it in range(100):
                y = net._tower_b1(x)
                bad_rep += int(not torch.equal(y, ref))
                k = 1 + it % bmax
                idx = torch.randperm(bmax, device="cuda")[:k]
                ys = net._tower_b1(x[idx].contiguous())
                bad_sub += int(not torch.equal(ys, ref[idx]))
                l, v = net(x[idx].contiguous())
                bad_fwd += int(not (torch.equal(l, l0[idx]) and torch.equal(v, v0[idx])))
        net.check_b1()
        print(size, "f32" if f32 else "split", "max batch", bmax, "repeat mismatches", bad_rep, "subset mismatches", bad_sub, "forward mismatches", bad_fwd, flush=True)
print("done")
