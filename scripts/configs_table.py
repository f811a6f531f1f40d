#!/usr/bin/env python3
"""scripts/configs_table.py -- one markdown row per bench.py log (the JSON line each run printed).
usage: configs_table.py "label=path/to/log" ...   (prints the table of profiles/rNN_all_configs.md)"""
import json, sys

print("| configuration | nodes/s | ms per ply | unique NN evals/s | games/hour measured | opening-phase ms per ply | roofline (dominant kernel) |")
print("|---|---|---|---|---|---|---|")
for arg in sys.argv[1:]:
    label, path = arg.rsplit("=", 1)
    lines = [l for l in open(path).read().splitlines() if l.startswith('{"metric')]
    if not lines:
        print(f"| {label} (`{path.split('/')[-1]}`) | no JSON line | | | | | |")
        continue
    d = json.loads(lines[-1])
    op = (d.get("opening_phase") or {}).get("ms_per_step", "")
    r = d.get("roofline") or {}
    roof = f"{r.get('kernel', '')}: {r.get('achieved')} {r.get('unit')} = {r.get('frac')}" if r else ""
    gph = d.get("games_per_hour_measured")
    print(f"| {label} (`{path.split('/')[-1]}`) | {d['value']:,.0f} | {d['ms_per_step']} | {d.get('unique_nn_evals_per_sec', 0):,.0f} | "
          f"{'' if gph is None else f'{gph:,.0f}'} | {op} | {roof} |")
