"""CPU tests: the C oracle against the golden traces produced by the UNMODIFIED reference
(tests/golden/generate_golden.py).  This is what pins oracle/bo_mcts.c + bo_codec.c to the
reference's mcts.py / utils.py / self_play.py (rows M1-M10, S1-S3, U1-U5 of SURVEY.md section 8a)."""
import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O

SEAM = G.SeamTable()
SEARCHES = G.load_searches()
GAMES = G.load_games()


def build_context(case):
    b = O.Board(case["fen"])
    trk = O.PyTracker()
    trk.add_board(b)
    for u in case["moves"]:
        b.push(u)
        trk.add_board(b)
    pos = b.positions()
    if case.get("uci_style"):
        hist = pos[-8:][-7:]
    else:
        hist = pos[max(0, len(pos) - 8):-1]
    return b, hist, trk


@pytest.mark.parametrize("entry", SEARCHES, ids=[e["case"]["name"] for e in SEARCHES])
def test_search_matches_reference_trace(entry):
    case, exp = entry["case"], entry["expect"]
    b, hist, trk = build_context(case)
    rng = np.random.RandomState(case["seed"])
    cfg = G.oracle_cfg(O, case["config"])
    fn = SEAM.eval_fn(case["scale"], case["salt"])
    if exp.get("raises"):
        with pytest.raises(ValueError):
            O.run_mcts(b, hist, trk, fn, rng, cfg)
        return
    r = O.run_mcts(b, hist, trk, fn, rng, cfg)
    assert O.move_to_uci(r["best"]) == exp["best"]
    nz = np.nonzero(r["pi"])[0]
    assert [[int(i), G.f32bits(r["pi"][i])] for i in nz] == exp["pi"]
    tree = O.canonical_tree(r["nodes"])
    got = {"/".join(k): list(v) for k, v in tree.items()}
    assert got == exp["tree"]
    assert r["n_batches"] == len(exp["batches"])
    assert r["n_batch_rows"] == sum(x[0] for x in exp["batches"])
    assert r["n_terminal_sims"] == case["config"]["num_simulations"] - r["n_batch_rows"]
    assert r["max_unique_in_batch"] <= max([x[1] for x in exp["batches"]] + [0])


@pytest.mark.parametrize("entry", GAMES, ids=[e["case"]["name"] for e in GAMES])
def test_self_play_game_matches_reference(entry):
    case, exp = entry["case"], entry["expect"]
    rng = np.random.RandomState(case["seed"])
    cfg = G.oracle_cfg(O, case["config"])
    g = O.self_play(SEAM.eval_fn(case["scale"], case["salt"]), rng, cfg, start_fen=case.get("fen", ""))
    assert g is not None
    assert [O.move_to_uci(m) for m in g["moves"]] == exp["moves"]
    assert len(g["records"]) == exp["n_records"]
    assert [z for _, _, z in g["records"]] == exp["z"]
    assert [bool(np.signbit(z)) for _, _, z in g["records"]] == exp["z_signbit"]
    for (state, pi, _), epi, esha in zip(g["records"], exp["pi"], exp["state_sha1"]):
        assert [[int(i), G.f32bits(pi[i])] for i in np.nonzero(pi)[0]] == epi
        assert G.planes_key(state) == esha


def test_codec_tables_match_reference():
    for e in G.load_codec():
        assert e["errors"] == 0
        b = O.Board(e["fen"])
        got = [[O.move_to_uci(m), O.move_to_index(m)] for m in b.legal_moves()]
        assert got == e["moves"]
        for uci, idx in e["moves"]:
            assert O.move_to_uci(O.index_to_move(idx, b.pos)) == uci


def test_numpy_pairwise_sum_restated_exactly():
    rng = np.random.RandomState(0)
    for n in (1, 7, 8, 9, 127, 128, 129, 1000, 4672, 4673):
        for _ in range(5):
            a = rng.rand(n).astype(np.float32) * np.float32(rng.choice([1e-4, 1.0, 37.0]))
            assert O.np_sum_f32(a).view(np.uint32) == a.sum().view(np.uint32), n
