#!/usr/bin/env python3
"""scripts/pmc_summary.py -- per-kernel averages of a rocprofv3 --pmc pass (counter_collection.csv) as a markdown table.
usage: pmc_summary.py <dir> [counter] [last_n]   (default FETCH_SIZE; the x2 column is the gfx950 correction of MI355X_MICROARCH.md;
last_n > 0: a second table over only the last_n dispatches of every kernel -- e.g. the launches a bench's roofline leg timed)"""
import csv, glob, os, sys, collections

d = sys.argv[1]
counter = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv under", d); sys.exit(1)
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
per = collections.defaultdict(dict)  # kernel -> dispatch id -> value
tot, n = collections.Counter(), collections.Counter()
seen = {}
for f in files:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"][:70]
        key = (r.get("Dispatch_Id"), k)
        tot[k] += float(r["Counter_Value"])
        did = int(r.get("Dispatch_Id") or 0)
        per[k][did] = per[k].get(did, 0.0) + float(r["Counter_Value"])
        if key not in seen:
            seen[key] = 1
            n[k] += 1
print(f"| kernel | dispatches | {counter} avg (KB) | x2 gfx950 correction (MB) |")
print("|---|---|---|---|")
for k in sorted(tot, key=lambda k: -tot[k]):
    avg = tot[k] / max(1, n[k])
    print(f"| {k} | {n[k]} | {avg:.1f} | {2 * avg / 1024:.2f} |")

if last_n > 0:
    print(f"\nlast {last_n} dispatches of each kernel:\n")
    print(f"| kernel | dispatches | {counter} avg (KB) | x2 gfx950 correction (MB) |")
    print("|---|---|---|---|")
    for k in sorted(per, key=lambda k: -tot[k]):
        ids = sorted(per[k])[-last_n:]
        avg = sum(per[k][i] for i in ids) / max(1, len(ids))
        print(f"| {k} | {len(ids)} | {avg:.1f} | {2 * avg / 1024:.2f} |")
