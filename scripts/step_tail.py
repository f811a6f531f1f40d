#!/usr/bin/env python3
"""scripts/step_tail.py -- LAB: what the slow plies of the bench's steady state have in common: per step its duration, the games that
finished in it and the evaluations beyond the expected ten per cohort.  usage: step_tail.py K [steps] [preroll]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from betaone_amd import engine as E
from betaone_amd.rollout import CohortRollout, Rollout

K = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200; PREROLL = int(sys.argv[3]) if len(sys.argv) > 3 else 640
G = 256
dev = torch.device("cuda:0")
E.load_hip_library()
_, net = bench.make_net("10x128", dev, "fp32", G // K)
kw = dict(num_simulations=800, mcts_batch_size=96, device=str(dev), use_graph=True, rng_mode="native", policy_kind="probs", max_game_moves=16384)
ro = CohortRollout(net, G, cohorts=K, **kw) if K > 1 else Rollout(net, G, **kw)
drv = bench.Driver(ro, 0, 1, None)
drv.preroll(PREROLL, G)
for _ in range(5):
    drv.step()
torch.cuda.synchronize()
rows = []
for _ in range(steps):
    f0, n0, h0 = ro.n_forward, drv.n_finished, ro.host_seconds
    t0 = time.perf_counter()
    drv.step()
    rows.append((time.perf_counter() - t0, ro.n_forward - f0, drv.n_finished - n0, ro.host_seconds - h0))
torch.cuda.synchronize()
a = np.array(rows)
ms = a[:, 0] * 1e3
print(f"K = {K}, preroll {PREROLL}, {steps} steps: mean {ms.mean():.3f} ms  p10/p50/p90/max {np.percentile(ms, 10):.2f} / {np.percentile(ms, 50):.2f} / {np.percentile(ms, 90):.2f} / {ms.max():.2f}")
extra = a[:, 1] - 10 * K
for name, m in (("no finished game, no extra evaluation", (a[:, 2] == 0) & (extra <= 0)), ("finished games only", (a[:, 2] > 0) & (extra <= 0)),
                ("extra evaluations only", (a[:, 2] == 0) & (extra > 0)), ("both", (a[:, 2] > 0) & (extra > 0))):
    if m.any():
        print(f"  {name:40s} {int(m.sum()):4d} steps  mean {ms[m].mean():.3f} ms  p50 {np.percentile(ms[m], 50):.2f}  p90 {np.percentile(ms[m], 90):.2f}   forwards {a[m, 1].mean():.2f}  finished {a[m, 2].mean():.2f}")
X = np.stack([np.ones(len(a)), np.maximum(extra, 0), a[:, 2]], axis=1)
coef, *_ = np.linalg.lstsq(X, ms, rcond=None)
print(f"  least squares: {coef[0]:.3f} ms + {coef[1]:.3f} ms per extra evaluation + {coef[2]:.3f} ms per finished game")
print("  mean ms per window of 20 steps:", [round(float(ms[i:i + 20].mean()), 3) for i in range(0, len(ms), 20)])
slow = np.argsort(-ms)[:8]
print("  slowest:", [(round(float(ms[i]), 2), int(a[i, 1]), int(a[i, 2])) for i in slow], "(ms, forwards, finished)")
ro.close()
