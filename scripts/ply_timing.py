import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from betaone_amd.rollout import Rollout
_, net = bench.make_net("10x128", torch.device("cuda:0"), "fp32", 256)
ro = Rollout(net, 256, num_simulations=800, mcts_batch_size=96, device="cuda:0", use_graph=True, rng_mode="native")
ro.start_games(list(range(256)), list(range(256)), list(range(256)))
import betaone_amd.rollout as R
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
wrap(ro, "_run_search_steps"); wrap(ro.eng, "selfplay_turn"); wrap(ro.eng, "selfplay_begin"); wrap(ro, "_play_ply_native")
for _ in range(5): ro.play_ply()
for k in T: T[k].clear()
t0 = time.perf_counter()
for _ in range(20): ro.play_ply()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 20
print("per ply %.1f us" % (tot * 1e6))
for k, v in T.items():
    if v: print("  %-22s n=%d mean %.1f us" % (k, len(v), np.mean(v) * 1e6))
