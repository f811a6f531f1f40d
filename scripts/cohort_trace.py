#!/usr/bin/env python3
"""scripts/cohort_trace.py -- what the device does in a bench run with cohorts, from rocprofv3's *_kernel_trace.csv:
share of wall time with at least one tower kernel running, with nothing running, mean kernel durations, and how the kernels
spread over hardware queues.  usage: cohort_trace.py <dir>   (second half of the trace = steady state)"""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or "")[:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
rows = rows[len(rows) // 2:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = t1 - t0


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)


tower = [(s, e) for s, e, n, q, st in rows if "tower" in n]
anyk = [(s, e) for s, e, n, q, st in rows]
plays = [r for r in rows if r[2].startswith("bo_k_play")]
print(f"span {span / 1e6:.1f} ms, {len(rows)} kernels, {len(plays)} bo_k_play launches")
print(f"tower running (union over streams): {union(tower) / span:.3f} of wall time; sum of tower durations / wall: {sum(e - s for s, e in tower) / span:.3f}")
print(f"device idle (nothing running): {1 - union(anyk) / span:.3f}")
by = {}
for s, e, n, q, st in rows:
    by.setdefault(n, []).append(e - s)
print("| kernel | launches | mean us | total / wall |")
print("|---|---|---|---|")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"| {n} | {len(v)} | {np.mean(v) / 1e3:.1f} | {sum(v) / span:.3f} |")
q = {}
for s, e, n, qq, st in rows:
    q.setdefault((qq, st), [0, 0])
    q[(qq, st)][0] += 1
    q[(qq, st)][1] += e - s
print("| queue, stream | kernels | busy / wall |")
print("|---|---|---|")
for k, (c, b) in sorted(q.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k} | {c} | {b / span:.3f} |")
