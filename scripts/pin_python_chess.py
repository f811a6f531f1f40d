#!/usr/bin/env python3
"""scripts/pin_python_chess.py -- lifts "parity unpinned" from the one layer the reference does not contain: python-chess.

The whole path hangs on python-chess's behaviour in three places -- the ORDER of board.legal_moves (stable-sort ties of mcts.py:58-62,
noise[i] <-> i-th legal move of mcts.py:190-197, first-max best move of mcts.py:279), is_game_over(claim_draw=True) / result
(mcts.py:152, utils.py:385-396) and _transposition_key (utils.py:78-103).  python-chess (chess==1.11.2, train_requirements.txt:2) is
not in /root/reference and not in this image (ModuleNotFoundError -- an absent package, not a refused action), so the oracle restates
it (oracle/bo_rules.c) and the golden fixtures were generated with oracle/shim/chess.py standing in for it.

Wherever the real package IS importable, this script
  1. dumps, with the real `chess`, the ordered legal-move lists, outcome(claim_draw=True) and _transposition_key-equality classes for
     the positions tests/test_oracle_rules.py and tests/golden/generate_golden.py use, and compares them with the oracle's;
  2. re-runs tests/golden/generate_golden.py's G2 / G5 cases with the real package on sys.path instead of the shim (into a temporary
     directory) and diffs the result against the committed fixtures: moves, pi bits, trees, z, state hashes.
Exit code 0 = everything agrees (or the package is absent: said so, nothing to compare); 1 = a difference, printed.

usage: python scripts/pin_python_chess.py [--skip-golden]
"""
from __future__ import annotations

import importlib.util
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def real_chess():
    """The real python-chess, never the shim: found on the default path only."""
    spec = importlib.util.find_spec("chess")
    if spec is None or (spec.origin or "").startswith(os.path.join(ROOT, "oracle", "shim")):
        return None
    import chess

    return chess if hasattr(chess, "Board") and hasattr(chess.Board, "_transposition_key") else None


def positions():
    """(fen, moves) of every position the rules tests and the golden cases start from."""
    import test_oracle_rules as T

    out = [(fen, []) for fen, _ in T.PERFT] + [(fen, list(mv)) for fen, mv, _ in T.TERMINATIONS]
    src = open(os.path.join(ROOT, "tests", "golden", "g2_searches.json")).read()
    for e in json.loads(src):
        out.append((e["case"]["fen"], list(e["case"]["moves"])))
    for f in ("g2_games.json", "g5_long_games.json"):
        for e in json.load(open(os.path.join(ROOT, "tests", "golden", f))):
            start = e["case"].get("fen") or "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
            mv = e["expect"].get("moves", [])
            for k in range(0, len(mv) + 1, max(1, len(mv) // 12)):  # a dozen positions along every recorded game
                out.append((start, mv[:k]))
    return out


def compare_rules(chess) -> int:
    from oracle import oracle as O

    bad = 0
    keys_real, keys_ours = {}, {}
    for i, (fen, moves) in enumerate(positions()):
        rb, ob = chess.Board(fen), O.Board(fen)
        for m in moves:
            rb.push(chess.Move.from_uci(m))
            ob.push(m)
        real = [m.uci() for m in rb.legal_moves]
        ours = [O.move_to_uci(m) for m in ob.legal_moves()]
        if real != ours:
            bad += 1
            print(f"ORDER differs at {fen} + {' '.join(moves)}:\n  python-chess {real}\n  oracle       {ours}")
        oc = rb.outcome(claim_draw=True)
        real_term = 0 if oc is None else {"CHECKMATE": 1, "INSUFFICIENT_MATERIAL": 2, "STALEMATE": 3, "SEVENTYFIVE_MOVES": 4, "FIVEFOLD_REPETITION": 5,
                                          "FIFTY_MOVES": 6, "THREEFOLD_REPETITION": 7}.get(oc.termination.name, -1)
        if real_term != ob.termination():
            bad += 1
            print(f"OUTCOME differs at {fen} + {' '.join(moves)}: python-chess {real_term}, oracle {ob.termination()}")
        keys_real.setdefault(rb._transposition_key(), []).append(i)
        k = ob.key()
        keys_ours.setdefault(tuple(getattr(k, f) for f, _ in k._fields_), []).append(i)
    if sorted(map(tuple, keys_real.values())) != sorted(map(tuple, keys_ours.values())):
        bad += 1
        print("_transposition_key puts these positions into different equality classes than the oracle's key")
    print(f"[rules] {len(positions())} positions compared with python-chess {chess.__version__}: {bad} difference(s)")
    return bad


def compare_golden() -> int:
    """generate_golden.py with the real package in front of the shim, into a scratch directory; then a field-by-field diff."""
    gen = os.path.join(ROOT, "tests", "golden", "generate_golden.py")
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, BETAONE_GOLDEN_REAL_CHESS="1", BETAONE_GOLDEN_OUT=tmp)
        r = subprocess.run([sys.executable, gen], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print("[golden] the generator failed with the real package:\n" + r.stdout[-2000:] + r.stderr[-2000:])
            return 1
        bad = 0
        for f in ("g2_searches.json", "g2_games.json", "g5_long_games.json", "g4_codec.json"):
            a = json.load(open(os.path.join(ROOT, "tests", "golden", f)))
            b = json.load(open(os.path.join(tmp, f)))
            for ea, eb in zip(a, b):
                if ea != eb:
                    bad += 1
                    name = ea.get("case", {}).get("name", ea.get("fen", "?"))
                    keys = [k for k in ea.get("expect", ea) if ea.get("expect", ea).get(k) != eb.get("expect", eb).get(k)]
                    print(f"[golden] {f}: {name} differs in {keys}")
        print(f"[golden] fixtures regenerated with the real python-chess: {bad} difference(s) against the committed ones")
        return bad


def main() -> int:
    chess = real_chess()
    if chess is None:
        print("python-chess is not installed here (chess==1.11.2 is pinned by the reference's train_requirements.txt:2; this image has no wheel "
              "and no network): nothing to compare.  Move ORDER, claim_draw and _transposition_key stay restated (oracle/bo_rules.c), pinned only "
              "by perft for the SET of moves and, for the start position, by the order python-chess's documentation prints "
              "(tests/test_oracle_rules.py).  Run this script wherever `pip install chess==1.11.2` is possible.")
        return 0
    bad = compare_rules(chess)
    if "--skip-golden" not in sys.argv:
        bad += compare_golden()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
