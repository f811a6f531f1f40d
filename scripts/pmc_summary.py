#!/usr/bin/env python3
"""scripts/pmc_summary.py -- per-kernel averages of a rocprofv3 --pmc pass (counter_collection.csv) as a markdown table.
usage: pmc_summary.py <dir> [counter] [last_n] [--json out.json]
  default counter FETCH_SIZE; last_n > 0: a second table over only the last_n dispatches of every kernel (e.g. the launches a bench's
  roofline leg timed).  Units per counter (MI355X_MICROARCH.md, HBM section):
    FETCH_SIZE, WRITE_SIZE  -- rocprofv3 reports KB; FETCH_SIZE reads HALF the bytes of wide streaming reads on gfx950, so a corrected
                               MB column (x 2) is printed for it -- and ONLY for it: WRITE_SIZE is exact;
    everything else         -- a plain event or cycle count, printed as such (SQ_* wave counters are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
                               and GRBM_GUI_ACTIVE cycles: the counter's own unit, no conversion here)."""
import csv, glob, json, os, sys, collections

argv = [a for a in sys.argv[1:]]
json_out = None
if "--json" in argv:
    i = argv.index("--json")
    json_out = argv[i + 1]
    del argv[i:i + 2]
d = argv[0]
counter = argv[1] if len(argv) > 1 else "FETCH_SIZE"
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv under", d); sys.exit(1)
last_n = int(argv[2]) if len(argv) > 2 else 0
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

BYTES = counter in ("FETCH_SIZE", "WRITE_SIZE")


def header():
    if counter == "FETCH_SIZE":
        return f"| kernel | dispatches | {counter} avg (KB, as reported) | bytes per dispatch, x2 gfx950 correction (MB) |\n|---|---|---|---|"
    if counter == "WRITE_SIZE":
        return f"| kernel | dispatches | {counter} avg (KB) | bytes per dispatch (MB, exact: no correction) |\n|---|---|---|---|"
    return f"| kernel | dispatches | {counter} avg per dispatch (count) |\n|---|---|---|"


def row(k, cnt, avg):
    if counter == "FETCH_SIZE":
        return f"| {k} | {cnt} | {avg:.1f} | {2 * avg / 1024:.2f} |"
    if counter == "WRITE_SIZE":
        return f"| {k} | {cnt} | {avg:.1f} | {avg / 1024:.2f} |"
    return f"| {k} | {cnt} | {avg:,.0f} |"


print(header())
summary = {}
for k in sorted(tot, key=lambda k: -tot[k]):
    avg = tot[k] / max(1, n[k])
    summary[k] = {"dispatches": n[k], "avg": avg}
    print(row(k, n[k], avg))

if last_n > 0:
    print(f"\nlast {last_n} dispatches of each kernel:\n")
    print(header())
    for k in sorted(per, key=lambda k: -tot[k]):
        ids = sorted(per[k])[-last_n:]
        avg = sum(per[k][i] for i in ids) / max(1, len(ids))
        summary[k]["last_n"] = {"dispatches": len(ids), "avg": avg}
        print(row(k, len(ids), avg))

if json_out:
    unit = "KB (x2 for bytes on gfx950)" if counter == "FETCH_SIZE" else "KB" if counter == "WRITE_SIZE" else "count"
    old = json.load(open(json_out)) if os.path.exists(json_out) else {}
    old[counter] = {"unit": unit, "kernels": summary}
    json.dump(old, open(json_out, "w"), indent=1)
