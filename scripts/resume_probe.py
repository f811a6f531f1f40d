#!/usr/bin/env python3
"""LAB: do games come out identical for different slot counts (24 games in 8 slots vs 3 of them in 3 slots)?  argv[1] = route override"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from betaone_amd import dropin
dropin.install()
import config, network, self_play
from betaone_amd import nn_tune
from betaone_amd.selfplay_main import game_seed
if len(sys.argv) > 1:
    forced = sys.argv[1]
    orig = nn_tune.kernel_route
    nn_tune.kernel_route = lambda f, b, d, p=None: forced if d == torch.float32 else orig(f, b, d, p)
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 3, 1, 64
config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 100, 96, 12
torch.manual_seed(0)
model = network.PolicyValueNet().to("cuda").eval()
ids = list(range(24))
a = self_play.run_self_play_games(model, ids, seeds=[game_seed(5, j) for j in ids], n_slots=8)
sub = [3, 11, 17]
b = self_play.run_self_play_games(model, sub, seeds=[game_seed(5, j) for j in sub], n_slots=3)
c = self_play.run_self_play_games(model, ids, seeds=[game_seed(5, j) for j in ids], n_slots=8)
for name, other, keys in (("8 slots again", c, ids), ("3 slots", b, sub)):
    for j in keys:
        x, y = a[j], other[j]
        if len(x) != len(y):
            print(name, "game", j, "lengths", len(x), len(y)); continue
        for k, (p, q) in enumerate(zip(x, y)):
            ds, dp, dz = not torch.equal(p[0], q[0]), not np.array_equal(p[1], q[1]), p[2] != q[2]
            if ds or dp or dz:
                print(name, "game", j, "ply", k, "state differs" if ds else "", "pi differs" if dp else "", "z differs" if dz else "",
                      "pi a", np.nonzero(p[1])[0], p[1][np.nonzero(p[1])[0]], "pi b", np.nonzero(q[1])[0], q[1][np.nonzero(q[1])[0]])
                break
print("done")
