#!/usr/bin/env python3
"""scripts/kernel_percentiles.py -- per-kernel launch-duration percentiles from rocprofv3's *_kernel_trace.csv
(the --stats table only has the mean; the step kernel's cost is its tail).  usage: kernel_percentiles.py <dir> [name-substring ...]
Only the second half of each kernel's launches is used (steady state of a bench run with pre-roll)."""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
want = sys.argv[2:] or ["bo_k_", "Cijk", "softmax", "copyBuffer", "fillBuffer"]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    print("no kernel_trace.csv under", d); sys.exit(1)
dur = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Name") or ""
        if not any(w in name for w in want):
            continue
        dur.setdefault(name[:80], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("| kernel | launches (2nd half) | mean us | p50 | p90 | p99 | max |")
print("|---|---|---|---|---|---|---|")
for name, v in sorted(dur.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    v.sort()
    a = np.array([x[1] for x in v[len(v) // 2:]], dtype=np.float64) / 1e3
    if len(a) == 0:
        continue
    print(f"| {name} | {len(a)} | {a.mean():.1f} | {np.percentile(a, 50):.1f} | {np.percentile(a, 90):.1f} | {np.percentile(a, 99):.1f} | {a.max():.1f} |")

# ---- gaps between consecutive launches (idle device time between the end of one kernel and the start of the next), by pair.
# Only meaningful for the single-stream part of the pipeline; overlapping launches (forked streams) show as negative and are skipped.
allk = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Name") or ""
        allk.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:48]))
allk.sort()
allk = allk[len(allk) // 2:]
gaps = {}
for (s0, e0, n0), (s1, e1, n1) in zip(allk, allk[1:]):
    g = s1 - e0
    if 0 <= g < 200000:  # > 0.2 ms: a host round trip, not a dependency gap
        gaps.setdefault((n0, n1), []).append(g)
print()
print("| previous kernel -> next kernel | pairs | mean gap us | p50 | p90 |")
print("|---|---|---|---|---|")
for (n0, n1), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    a = np.array(v, dtype=np.float64) / 1e3
    print(f"| {n0} -> {n1} | {len(a)} | {a.mean():.1f} | {np.percentile(a, 50):.1f} | {np.percentile(a, 90):.1f} |")

# ---- where a ply goes: the span between two bo_k_play launches, by kernel and idle (union of kernel intervals) ----
plays = [k for k in allk if k[2].startswith("bo_k_play")]
if len(plays) > 2:
    t0, t1 = plays[0][0], plays[-1][0]
    n_plies = len(plays) - 1
    inside = [k for k in allk if t0 <= k[0] < t1]
    by = {}
    for s, e, n in inside:
        by[n] = by.get(n, 0) + (e - s)
    busy, cur_s, cur_e = 0, None, None
    for s, e, n in inside:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    span = t1 - t0
    print()
    print(f"| per ply ({n_plies} plies, {span / n_plies / 1e3:.1f} us each under the profiler) | us | share |")
    print("|---|---|---|")
    for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:10]:
        print(f"| {n} | {v / n_plies / 1e3:.1f} | {v / span:.3f} |")
    print(f"| device idle (no kernel or copy running) | {(span - busy) / n_plies / 1e3:.1f} | {(span - busy) / span:.3f} |")
