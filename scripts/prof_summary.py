#!/usr/bin/env python3
"""scripts/prof_summary.py -- turn rocprofv3's *_kernel_stats.csv into the markdown table kept under profiles/.
usage: prof_summary.py <kernel_stats.csv> [top_n]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| kernel | calls | avg us | total ms | % of GPU time |")
print("|---|---|---|---|---|")
for r in rows[:top]:
    t = float(r["TotalDurationNs"])
    print(f"| {r['Name'][:110]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {t / 1e6:.2f} | {100 * t / tot:.1f} |")
print(f"\ntotal GPU kernel time {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
