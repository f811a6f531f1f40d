#!/usr/bin/env python3
"""scripts/step_profile.py -- where bo_k_step spends its cycles in steady-state self-play (per-phase s_memtime counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
from betaone_amd.rollout import Rollout

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 400
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # > 1: only game-steps longer than that many cycles
_, net = bench.make_net("10x128", torch.device("cuda:0"), "fp32", 256)
ro = Rollout(net, 256, num_simulations=800, mcts_batch_size=96, device="cuda:0", use_graph=False, rng_mode="native",
             policy_kind=os.environ.get("BO_PROFILE_KIND", "logits"))  # logits (+ BETAONE_STEP_TAIL=0 / 1) or probs
print("policy_kind", os.environ.get("BO_PROFILE_KIND", "logits"), "step_tail", ro.step_tail)
ro.start_games(list(range(256)), list(range(256)), list(range(256)))
nid = [256]
def refill(_s):
    nid[0] += 1
    return nid[0], nid[0], None
for _ in range(warm):
    ro.play_ply(refill=refill)
ro.eng.profile(enable=thr, read=False)
for _ in range(steps):
    ro.play_ply(refill=refill)
p = ro.eng.profile(enable=0).astype(np.float64)
names = ["apply", "select", "first-visit", "terminal-backup", "encode", "flush", "total", "steps"]
if thr > 1:
    n = p[:, 7].sum()
    print(f"game-steps longer than {thr} cycles: {int(n)} of {256 * steps * 11} (~{n / (steps * 11):.2f} per launch)")
    tot = p.sum(axis=0)
    for i, nm in enumerate(names[:7]):
        print(f"  {nm:16s} {tot[i] / max(n, 1):10.0f} cycles per slow step")
    print(f"  loop iterations per slow step {tot[8] / max(n, 1):.1f}, first visits {tot[9] / max(n, 1):.1f}")
    print(f"  terminal iterations per slow step: burst calls {tot[10] / max(n, 1):.1f} applying {tot[11] / max(n, 1):.1f} simulations, "
          f"general-path simulations {tot[12] / max(n, 1):.1f}; cycles before the burst (path check) {tot[13] / max(n, 1):.0f}, inside burst / general backup {tot[14] / max(n, 1):.0f}")
    print(f"  switches between the burst's two register-resident paths: {tot[15] / max(n, 1):.1f} per slow step ({int(tot[15])} in all)")
    sys.exit(0)
per_step = p[:, :7] / np.maximum(p[:, 7:8], 1)
print("cycles per step per game (100 MHz s_memtime ticks? shader clock): mean over games / max over games")
for i, n in enumerate(names[:7]):
    print(f"  {n:16s} mean {per_step[:, i].mean():10.0f}   max {per_step[:, i].max():10.0f}")
st = ro.eng.status()
print("term_sims per game-step:", st["term_sims"].sum() / max(1, p[:, 7].sum()), " evals:", st["evals"].sum())
print("burst calls", int(p[:, 10].sum()), "simulations applied in bursts", int(p[:, 11].sum()), "switches between the two register-resident paths", int(p[:, 15].sum()))
