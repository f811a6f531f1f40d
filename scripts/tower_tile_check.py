#!/usr/bin/env python3
"""scripts/tower_tile_check.py -- the two tilings of the 128-filter split-precision tower side by side on one box: accuracy against a float64
evaluation of the same net and the latency of a lone 64- / 256-board launch (event pairs around 50 back-to-back launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from betaone_amd import dropin
dropin.install()
import config, network
from betaone_amd.fused_net import FusedPolicyValueNet
from fake_model import hash_init_

config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
net = hash_init_(network.PolicyValueNet().eval()).cuda()
ref = network.PolicyValueNet().eval()
ref.load_state_dict(net.state_dict())
ref = ref.double().cuda()
torch.manual_seed(1)
x = (torch.rand(256, 120, 8, 8, device="cuda") < 0.15).float()
with torch.no_grad():
    l64, v64 = ref(x.double())
for tile in ("32", "16"):
    os.environ["BETAONE_SPLIT_TILE"] = tile
    f = FusedPolicyValueNet(net, conv="tower_split").cuda()
    with torch.no_grad():
        l, v = f(x)
        torch.cuda.synchronize()
        err_l, err_v = (l.double() - l64).abs().max().item(), (v.double() - v64).abs().max().item()
        for B in (64, 256):
            xb = x[:B].contiguous()
            for _ in range(5):
                f._tower_forward(xb, heads=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                f._tower_forward(xb, heads=True)
            e1.record()
            torch.cuda.synchronize()
            print(f"tile {tile}: {B:3d} boards {e0.elapsed_time(e1) * 1000 / 50:7.1f} us per launch (back to back)")
    f.check_overflow()
    print(f"tile {tile}: max |logit error| vs float64 {err_l:.2e}, |value error| {err_v:.2e}")
