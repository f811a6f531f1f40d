#!/usr/bin/env python3
"""scripts/cohort_timeline.py -- LAB: where a cohort's iteration goes when K cohorts share the chip, by the device's own clock: a
one-thread stamp kernel (bo_debug_stamp) sits in every cohort's captured graphs in front of the forward (1), behind it (2) and behind
the tree step (3).  usage: cohort_timeline.py K [games] [steps] [preroll plies]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from betaone_amd import engine as E
from betaone_amd.rollout import CohortRollout, Rollout

K = int(sys.argv[1]);  G = int(sys.argv[2]) if len(sys.argv) > 2 else 256; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
PREROLL = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda:0")
E.load_hip_library()
_, net = bench.make_net("10x128", dev, "fp32", G // K)
Rollout.STAMP_RING = torch.zeros(1 + 2 * Rollout.STAMP_CAP, dtype=torch.int64, device=dev)
kw = dict(num_simulations=800, mcts_batch_size=96, device=str(dev), use_graph=True, rng_mode="native", policy_kind="probs")
ro = CohortRollout(net, G, cohorts=K, **kw) if K > 1 else Rollout(net, G, **kw)
drv = bench.Driver(ro, 0, 1, None)
drv.preroll(PREROLL, G)
for _ in range(5):
    drv.step()
torch.cuda.synchronize()
Rollout.STAMP_RING.zero_()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    drv.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ring = Rollout.STAMP_RING.cpu().numpy().astype(np.uint64)
n = int(min(ring[0], Rollout.STAMP_CAP))
tags, ts = ring[1:1 + 2 * n:2].astype(np.int64), ring[2:2 + 2 * n:2].astype(np.float64)
khz = ctypes.c_int32(0); E.load_hip_library().bo_device_wall_clock_khz(0, ctypes.byref(khz))
ts = ts / (khz.value or 100000) * 1e3  # us
MASKS = getattr(ro, "cu_masks", "-")
print(f"K = {K} (cu masks {MASKS}), preroll {PREROLL}, {G} games, {steps} steps: {dt / steps * 1e3:.3f} ms per step, {n} stamps")
ids = sorted(set(int(t) // 16 for t in tags))
fw, st, gap = [], [], []
for cid in ids:
    m = (tags // 16) == cid
    ph, tt = tags[m] % 16, ts[m]
    o = np.argsort(tt, kind="stable"); ph, tt = ph[o], tt[o]
    for i in range(len(ph) - 1):
        d = tt[i + 1] - tt[i]
        if ph[i] == 1 and ph[i + 1] == 2: fw.append(d)
        elif ph[i] == 2 and ph[i + 1] == 3: st.append(d)
        elif ph[i] == 3 and ph[i + 1] == 1 and d < 2000: gap.append(d)
pc = lambda v: " / ".join(f"{x:.0f}" for x in np.percentile(v, [10, 50, 90])) if len(v) else "-"
print(f"forward (tower + 2 head kernels), stamp 1 -> 2: mean {np.mean(fw):.1f} us  p10/p50/p90 {pc(fw)}   n {len(fw)}")
print(f"tree step, stamp 2 -> 3:                        mean {np.mean(st):.1f} us  p10/p50/p90 {pc(st)}   n {len(st)}")
print(f"step -> next forward (graph node to node), 3 -> 1: mean {np.mean(gap):.1f} us  p10/p50/p90 {pc(gap)}   n {len(gap)}")
ro.close()
