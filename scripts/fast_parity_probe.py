#!/usr/bin/env python3
"""scripts/fast_parity_probe.py -- diagnostic (GPU box): the fast-mode kernels on the product library against the wave-emulator
build of the SAME sources, in lock step, one engine step at a time; reports the first step at which the trees differ, and
which select-kernel variants pass tests/fast_reference.py.  Test infrastructure, not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import test_fast_mode_emu as T  # noqa: E402
from betaone_amd import engine as E  # noqa: E402
from engine_harness import Buf, emu_call  # noqa: E402
from fast_reference import canonical_from_engine  # noqa: E402


def lockstep(fen, moves, sims, L, opts, max_steps=400):
    kw = dict(num_simulations=sims, dirichlet_alpha=0.1, fast=True, leaves_per_step=L, max_plies=256)
    engs = {"emu": emu_call(E.Engine, 1, **kw), "hip": E.Engine(1, **kw)}
    fn = T.softmax_eval(7)
    bufs = {}
    for b, eng in engs.items():
        eng.fast_options(**opts)
        eng.reset([0], [fen], [" ".join(moves) or None])
        nl, term, _ = eng.root_info()
        noise = np.zeros((1, E.MAX_LEGAL))
        noise[0, :nl[0]] = np.random.RandomState(3).dirichlet([0.1] * int(nl[0]))
        bufs[b] = (Buf(b, (L, 120, 8, 8)), Buf(b, (L, E.NUM_ACTIONS)), Buf(b, (L,)))
        eng.search_begin([1], noise, bufs[b][0].ptr)
    kind = E.POLICY_NONE
    for step in range(max_steps):
        trees = {}
        for b, eng in engs.items():
            nn_in, pol, val = bufs[b]
            eng.step(pol.ptr, val.ptr, kind, nn_in.ptr)
            trees[b] = canonical_from_engine(eng.debug_tree(0), E.move_to_uci)
        if trees["emu"] != trees["hip"]:
            keys = sorted(set(trees["emu"]) | set(trees["hip"]))
            diff = [(k, trees["emu"].get(k), trees["hip"].get(k)) for k in keys if trees["emu"].get(k) != trees["hip"].get(k)]
            print(f"  DIVERGED at step {step}: {len(diff)} entries differ; first 6 (path, emu, hip):")
            for d in diff[:6]:
                print("   ", d)
            return False
        planes = {b: bufs[b][0].numpy().copy() for b in engs}
        dbg = {b: engs[b].debug_fast(0) for b in engs}
        same_ctl = all(np.array_equal(dbg["emu"][0][k], dbg["hip"][0][k]) for k in dbg["emu"][0])
        if not same_ctl or not np.array_equal(planes["emu"], planes["hip"]):
            bad = [r for r in range(L) if not np.array_equal(planes["emu"][r], planes["hip"][r])]
            print(f"  step {step}: trees equal; NN input rows that differ: {bad}; control blocks equal: {same_ctl}")
            for b in ("emu", "hip"):
                c, paths = dbg[b]
                print(f"    {b}: n_rows {c['n_rows']} n_step {c['n_step']} top {c['top']} row_slot {c['row_slot'].tolist()} row_plink {c['row_plink'].tolist()} "
                      f"row_sim {c['row_sim'].tolist()} sim_row {c['sim_row'].tolist()} sim_plen {c['sim_plen'].tolist()} row_term {c['row_term'].tolist()} "
                      f"row_nlegal {c['row_nlegal'].tolist()} paths {[paths[s, :max(1, int(c['sim_plen'][s]))].tolist() for s in range(min(L, 4))]}")
        running = [engs[b].poll()[0] for b in engs]
        if running[0] != running[1]:
            print(f"  step {step}: running differs {running}")
            return False
        if not running[0]:
            print(f"  ok: {step + 1} steps, trees identical after every step")
            return True
        for b in engs:
            p, v = fn(bufs[b][0].numpy())
            bufs[b][1].set(p); bufs[b][2].set(v)
        kind = E.POLICY_PROBS
    return True


if __name__ == "__main__":
    print("== lock step emu vs hip, one game ==")
    for L, ut, fl in [(4, 4, 0), (4, 2, 0), (4, 4, 2), (8, 4, 0), (33, 2, 0)]:
        print(f"L={L} games_per_halfwave={ut} flags={fl}")
        lockstep(T.CASES[0][0], T.CASES[0][1], 64, L, dict(games_per_halfwave=ut, select_flags=fl))
    print("== variants against the NumPy restatement, a different position in every slot ==")
    for L, sims in [(4, 60), (8, 64), (33, 99)]:
        for ut in (4, 2):
            for fl in (0, 3, 7):
                try:
                    T.check_multi("hip", L, sims, dict(games_per_halfwave=ut, select_flags=fl))
                    res = "ok"
                except AssertionError as ex:
                    res = "FAIL " + str(ex)[:80].replace("\n", " ")
                print(f"L={L} ut={ut} flags={fl}: {res}", flush=True)
