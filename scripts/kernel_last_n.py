#!/usr/bin/env python3
"""scripts/kernel_last_n.py -- mean / p50 duration of the LAST n dispatches of the kernels matching a regex in a rocprofv3 --kernel-trace
output directory (the launches a bench's roofline leg timed at the end of its run), and of all of them.
usage: kernel_last_n.py <dir> <regex> <n>"""
import csv, glob, os, re, sys
import statistics as st

d, rx, n = sys.argv[1], re.compile(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60]))
rows.sort()
if not rows:
    print("no dispatch matches", sys.argv[2]); sys.exit(1)
dur = [x[1] for x in rows]
print(f"| kernel | dispatches | mean us (all) | mean us (last {n}) | p50 us (last {n}) | min / max us (last {n}) |")
print("|---|---|---|---|---|---|")
last = dur[-n:]
print(f"| {rows[-1][2]} | {len(dur)} | {st.mean(dur):.2f} | {st.mean(last):.2f} | {st.median(last):.2f} | {min(last):.2f} / {max(last):.2f} |")
