#!/usr/bin/env python3
"""LAB: conv='tower_b1' on rotating inputs, every output compared bitwise with that input's first result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet

size = tuple(int(v) for v in sys.argv[1:4]); B = int(sys.argv[4]); iters = int(sys.argv[5]); graph = sys.argv[6] == "graph"
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = size
torch.manual_seed(0)
plain = network.PolicyValueNet().cuda().eval()
net = FusedPolicyValueNet(plain, conv="tower_b1").cuda()
per = FusedPolicyValueNet(plain, conv="mfma_small").cuda()
xs = [torch.rand(B, 120, 8, 8, device="cuda") for _ in range(6)]
with torch.no_grad():
    refs = []
    for x in xs:
        r = net._tower_b1(x).clone(); torch.cuda.synchronize()
        for _ in range(3):
            assert torch.equal(net._tower_b1(x), r)
        refs.append(r)
        print("vs per-layer:", (r - per._tower_small(x)).abs().max().item())
    xg = xs[0].clone()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ys = [net._tower_b1(xg) for _ in range(3)]   # three launches back to back in one graph
    bad = 0
    for it in range(iters):
        i = it % 6
        if graph:
            xg.copy_(xs[i]); g.replay(); outs = ys
        else:
            outs = [net._tower_b1(xs[i])]
        for k, y in enumerate(outs):
            if not torch.equal(y, refs[i]):
                bad += 1
                if bad <= 4:
                    d = (y != refs[i])
                    nz = d.nonzero()
                    print(f"iteration {it} launch {k}: {int(d.sum())} values differ; boards {sorted(set(nz[:,0].tolist()))} channels {nz[:,1].min().item()}..{nz[:,1].max().item()} "
                          f"rows {nz[:,2].min().item()}..{nz[:,2].max().item()}; max |diff| {(y - refs[i]).abs().max().item():.3e}; equals another input's result: "
                          f"{[j for j in range(6) if torch.equal(y, refs[j])]}")
    net.check_b1()
    print(size, "B", B, "graph" if graph else "eager", "mismatching launches:", bad, "of", iters * (3 if graph else 1), flush=True)
