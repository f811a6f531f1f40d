"""tests/engine_cases.py -- backend-independent bodies of the engine parity tests (emu on CPU, hip on GPU)."""
from __future__ import annotations

import numpy as np

import golden_util as G
from betaone_amd import engine as E
from engine_harness import Searcher, canonical_tree, dense_pi, make_engine, Buf
from oracle import oracle as O

_seam = None


def seam():
    global _seam
    if _seam is None:
        _seam = G.SeamTable()
    return _seam


SEARCHES = {e["case"]["name"]: e for e in G.load_searches()}
GAMES = {e["case"]["name"]: e for e in G.load_games()}
SEARCH_NAMES = list(SEARCHES)
GAME_NAMES = list(GAMES)


def to_bo_position(p: O.Pos, ep_key: int = -2) -> E.BoPosition:
    b = E.BoPosition()
    for i, v in enumerate([p.pawns, p.knights, p.bishops, p.rooks, p.queens, p.kings, p.occ[1], p.occ[0]]):
        b.bb[i] = v
    b.turn = p.turn
    c = p.castling
    b.castling = (1 if c & (1 << 7) else 0) | (2 if c & 1 else 0) | (4 if c & (1 << 63) else 0) | (8 if c & (1 << 56) else 0)
    b.ep_square, b.ep_key = p.ep_square, ep_key
    b.halfmove_clock, b.fullmove_number = p.halfmove_clock, p.fullmove_number
    return b


def key_to_bo_position(k: O.Key) -> E.BoPosition:
    b = E.BoPosition()
    for i, v in enumerate([k.pawns, k.knights, k.bishops, k.rooks, k.queens, k.kings, k.occ_w, k.occ_b]):
        b.bb[i] = v
    b.turn = k.turn
    c = k.castling
    b.castling = (1 if c & (1 << 7) else 0) | (2 if c & 1 else 0) | (4 if c & (1 << 63) else 0) | (8 if c & (1 << 56) else 0)
    b.ep_square, b.ep_key = k.ep, k.ep
    b.halfmove_clock, b.fullmove_number = 0, 1
    return b


# ---- move generator ------------------------------------------------------------------------------------
def _oracle_moves(board: O.Board):
    return [O.move_to_uci(m) for m in board.legal_moves()]


def check_movegen_random_positions(backend, n_games=12, max_plies=60, seed=0):
    rng = np.random.RandomState(seed)
    boards, positions = [], []
    for _ in range(n_games):
        b = O.Board()
        for _ply in range(max_plies):
            mv = b.legal_moves()
            if not mv or b.termination() in (1, 2, 3):
                break
            positions.append((b.pos.copy(), _oracle_moves(b)))
            b.push(mv[rng.randint(len(mv))])
    eng = make_engine(backend, 1, dict(num_simulations=1))
    got, _ = eng.movegen([to_bo_position(p) for p, _ in positions])
    bad = 0
    for (p, exp), g in zip(positions, got):
        if [E.move_to_uci(m) for m in g] != exp:
            bad += 1
            if bad < 3:
                buf = O.C.create_string_buffer(128)
                O.lib().bo_pos_to_fen(O.C.byref(p), buf, 128)
                print("MISMATCH", buf.value.decode(), "\n  exp", exp, "\n  got", [E.move_to_uci(m) for m in g])
    assert bad == 0, f"{bad} of {len(positions)} positions differ"


SPECIAL_FENS = [
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R b KQkq - 0 1",
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",
    "r2q1rk1/pP1p2pp/Q4n2/bbp1p3/Np6/1B3NBn/pPPP1PPP/R3K2R b KQ - 0 1",
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
    "rnbqkbnr/ppp2ppp/8/1B1pp3/4P3/8/PPPP1PPP/RNBQK1NR b KQkq - 1 3",
    "rnbqkbnr/ppp1pppp/8/8/3pP3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 3",
    "8/8/8/2k5/3Pp3/8/8/4K3 b - d3 0 1",            # ep capture available while in check by the pushed pawn
    "8/8/8/8/k2Pp2Q/8/8/4K3 b - d3 0 1",            # ep capture illegal: discovers the queen on the rank
    "4k3/8/8/8/8/8/8/R3K2R w KQ - 0 1",
    "4k3/8/8/8/8/8/8/R3K1r1 w Q - 0 1",             # g1 attacked, queenside still fine
    "4k3/P6P/8/8/8/8/p6p/4K3 b - - 0 1",
    "k7/8/1K6/8/8/8/8/7R w - - 0 1",
    "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 1",
    "R6k/8/7K/8/8/8/8/8 b - - 0 1",                 # checkmate: no moves
    "7k/5Q2/6K1/8/8/8/8/8 b - - 0 1",               # stalemate
    "3rk3/8/8/8/8/8/3B4/3K4 w - - 0 1",             # pinned bishop
    "4k3/8/8/8/7b/8/5P2/4K3 w - - 0 1",             # pinned pawn (diagonal)
    "4k3/4r3/8/8/8/8/4P3/4K3 w - - 0 1",            # pawn pinned on the file: pushes allowed
    "4k3/8/8/8/8/2n5/8/R3K2R w KQ - 0 1",           # double check is impossible here; knight checks
    "2r1k3/8/8/8/8/8/8/R3K2R w KQ - 0 1",           # c1 attacked: no queenside castling
    "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1",  # 218 legal moves: the known maximum
    "8/8/8/8/8/8/8/K1k4R w - - 0 1",                 # rook check along the rank from far away
    "4k3/8/8/8/1b6/8/3P4/4K3 w - - 0 1",            # pawn pinned diagonally: no push, no capture
    "8/8/8/KPp4r/8/8/8/7k w - c6 0 1",              # ep capture would expose the king on the rank
]


def check_movegen_special(backend):
    boards = [O.Board(f) for f in SPECIAL_FENS]
    eng = make_engine(backend, 1, dict(num_simulations=1))
    got, chk = eng.movegen([to_bo_position(b.pos) for b in boards])
    for b, g, c, fen in zip(boards, got, chk, SPECIAL_FENS):
        assert [E.move_to_uci(m) for m in g] == _oracle_moves(b), fen
        assert bool(c) == bool(O.lib().bo_is_check(O.C.byref(b.pos))), fen


# ---- single searches against the golden traces --------------------------------------------------------------
def _history_and_tracker(case):
    """Explicit context for uci.py-style calls (history contains the root)."""
    b = O.Board(case["fen"])
    keys = [b.key()]
    ks = [Kcopy(keys[0])]
    for u in case["moves"]:
        b.push(u)
        ks.append(Kcopy(b.key()))
    pos = b.positions()
    hist = pos[-8:][-7:]
    return [to_bo_position(p) for p in hist], [(key_to_bo_position(k), 1) for k in ks]


def Kcopy(k: O.Key) -> O.Key:
    c = O.Key()
    O.C.memmove(O.C.byref(c), O.C.byref(k), O.C.sizeof(O.Key))
    return c


def check_golden_search(backend, name):
    entry = SEARCHES[name]
    case, exp = entry["case"], entry["expect"]
    eng = make_engine(backend, 1, case["config"])
    moves = " ".join(case["moves"]) or None
    if case.get("uci_style"):
        hist, trk = _history_and_tracker(case)
        eng.reset_ex([0], [case["fen"]], [moves], [hist], [trk])
    else:
        eng.reset([0], [case["fen"]], [moves])
    s = Searcher(backend, eng)
    rng = np.random.RandomState(case["seed"])
    res = s.search([1], [seam().eval_fn(case["scale"], case["salt"])], [rng], case["config"].get("dirichlet_alpha", 0.1))
    if exp.get("raises"):
        assert res["best_idx"][0] == -1
        return
    assert E.move_to_uci(int(res["best_move"][0])) == exp["best"]
    pi = dense_pi(res, 0)
    assert [[int(i), G.f32bits(pi[i])] for i in np.nonzero(pi)[0]] == exp["pi"]
    assert canonical_tree(eng.debug_tree(0)) == exp["tree"]
    st = eng.status()
    rows = sum(b[0] for b in exp["batches"])
    assert st["term_sims"][0] == case["config"]["num_simulations"] - rows
    assert st["flushes"][0] == len(exp["batches"])


# ---- whole games ------------------------------------------------------------------------------------------
def play_games(backend, eng, eval_fns, rngs, alpha, max_game_moves, temperature=(30, 1.0, 0.1), max_plies=None):
    """The host side of run_self_play_game for all slots at once (self_play.py:101-216)."""
    G_ = eng.G
    s = Searcher(backend, eng)
    pis = [[] for _ in range(G_)]
    active = np.ones(G_, dtype=bool)
    counts = np.zeros(G_, dtype=int)
    while True:
        nl, term, ply = eng.root_info()
        for g in range(G_):
            if active[g] and (term[g] != 0 or counts[g] >= max_game_moves):
                active[g] = False
        if not active.any():
            break
        # fullmove number of every root: derive from ply and the start position via export (cheap here)
        res = s.search(active.astype(np.int32), eval_fns, rngs, alpha)
        actions = np.full(G_, -1, dtype=np.int32)
        for g in range(G_):
            if not active[g]:
                continue
            pi = dense_pi(res, g)
            pis[g].append(pi)
            pos, _ = eng.export_game(g)
            th, ti, tf = temperature
            actions[g] = O.select_move_with_temperature(pi, pos[-1].fullmove_number, rngs[g], th, ti, tf)
            counts[g] += 1
        eng.play(actions)
    eng.check_status()
    out = []
    for g in range(G_):
        pos, mv = eng.export_game(g)
        _, term, _ = eng.root_info()
        outcome = 1.0 if term[g] == 1 else 0.0
        n = len(pis[g])
        buf = Buf(backend, (max(1, n), 120, 8, 8))
        if n:
            eng.encode_game(g, 0, n, buf.ptr)
        states = buf.numpy()[:n].copy()
        z = [outcome if pos[i].turn == 1 else -outcome for i in range(n)]
        out.append(dict(moves=[E.move_to_uci(m) for m in mv], pis=pis[g], states=states, z=z, outcome=outcome))
    return out


def check_golden_game(backend, name):
    entry = GAMES[name]
    case, exp = entry["case"], entry["expect"]
    cfg = case["config"]
    eng = make_engine(backend, 1, cfg)
    eng.reset([0], [case.get("fen")], [None])
    rng = np.random.RandomState(case["seed"])
    g = play_games(backend, eng, [seam().eval_fn(case["scale"], case["salt"])], [rng], cfg.get("dirichlet_alpha", 0.1),
                   cfg.get("max_game_moves", 16384))[0]
    assert g["moves"] == exp["moves"]
    assert len(g["pis"]) == exp["n_records"]
    assert g["z"] == exp["z"]
    assert [bool(np.signbit(z)) for z in g["z"]] == exp["z_signbit"]
    for pi, st, epi, esha in zip(g["pis"], g["states"], exp["pi"], exp["state_sha1"]):
        assert [[int(i), G.f32bits(pi[i])] for i in np.nonzero(pi)[0]] == epi
        assert G.planes_key(st) == esha


# ---- many games in lock step vs the oracle (fresh inputs, not fixtures) ----------------------------------------
def check_multi_game_vs_oracle(backend, n_games=5, plies=6, sims=60, batch=16, scale=6.0, **search_cfg):
    """search_cfg: cpuct / widen_coeff / dirichlet_alpha / dirichlet_eps overrides (engine and oracle get the same)."""
    from fake_model import fake_logits_values

    def eval_fn_for(salt):
        def fn(planes):
            logits, v = fake_logits_values(planes, scale, salt)
            x = logits.astype(np.float64)
            e = np.exp(x - x.max(axis=1, keepdims=True))
            return (e / e.sum(axis=1, keepdims=True)).astype(np.float32), v
        return fn

    cfg = dict(num_simulations=sims, batch_size=batch, max_game_moves=plies, **search_cfg)
    eng = make_engine(backend, n_games, cfg)
    eng.reset(list(range(n_games)))
    fns = [eval_fn_for(1000 + g) for g in range(n_games)]
    got = play_games(backend, eng, fns, [np.random.RandomState(g) for g in range(n_games)], cfg.get("dirichlet_alpha", 0.1), plies)
    ocfg = O.default_config(**cfg)
    for g in range(n_games):
        ref = O.self_play(fns[g], np.random.RandomState(g), ocfg)
        assert [O.move_to_uci(m) for m in ref["moves"]] == got[g]["moves"], g
        for (st, pi, z), gpi, gst, gz in zip(ref["records"], got[g]["pis"], got[g]["states"], got[g]["z"]):
            assert np.array_equal(pi.view(np.uint32), gpi.view(np.uint32))
            assert np.array_equal(st, gst)
            assert z == gz


def check_full_games_vs_oracle(backend, n_games=4, sims=12, batch=8, scale=4.0, max_game_moves=400):
    from fake_model import fake_logits_values

    def eval_fn_for(salt):
        def fn(planes):
            logits, v = fake_logits_values(planes, scale, salt)
            x = logits.astype(np.float64)
            e = np.exp(x - x.max(axis=1, keepdims=True))
            return (e / e.sum(axis=1, keepdims=True)).astype(np.float32), v
        return fn

    cfg = dict(num_simulations=sims, batch_size=batch, max_game_moves=max_game_moves)
    eng = make_engine(backend, n_games, cfg, max_plies=max_game_moves + 8)
    eng.reset(list(range(n_games)))
    fns = [eval_fn_for(500 + g) for g in range(n_games)]
    got = play_games(backend, eng, fns, [np.random.RandomState(100 + g) for g in range(n_games)], 0.1, max_game_moves)
    ocfg = O.default_config(**cfg)
    lengths = []
    for g in range(n_games):
        ref = O.self_play(fns[g], np.random.RandomState(100 + g), ocfg)
        assert [O.move_to_uci(m) for m in ref["moves"]] == got[g]["moves"], g
        assert ref["outcome"] == got[g]["outcome"]
        for (st, pi, z), gpi, gst, gz in zip(ref["records"], got[g]["pis"], got[g]["states"], got[g]["z"]):
            assert np.array_equal(pi.view(np.uint32), gpi.view(np.uint32))
            assert np.array_equal(st, gst)
            assert z == gz and np.signbit(z) == np.signbit(gz)
        lengths.append((len(ref["moves"]), ref["termination"]))
    assert all(l > 20 for l, _ in lengths), lengths
    return lengths


def check_edge_cases(backend):
    """Maximum sizes and error paths of the C ABI."""
    # 218 children in the FAST mode (full-width expansion) and a reference-mode search on the same position
    fen = "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1"
    import test_fast_mode_emu as T
    eng, _, _ = T.run_engine_search(backend, fen, [], 64, 8, T.softmax_eval(5), seed=2)
    t = eng.debug_tree(0)
    assert t[0]["n_children"] == 218 and sum(k["n"] for k in t[1:219]) == 64
    eng = make_engine(backend, 1, dict(num_simulations=100))
    eng.reset([0], [fen], [None])
    nl, term, _ = eng.root_info()
    assert nl[0] == 218 and term[0] == 0
    s = Searcher(backend, eng)
    res = s.search([1], [T.softmax_eval(5)], [np.random.RandomState(0)], 0.1)
    assert int(res["total"][0]) == 100
    # unsupported configurations are refused at create time, with a message
    mk = (lambda *a, **k: __import__("engine_harness").emu_call(E.Engine, *a, **k)) if backend == "emu" else E.Engine
    for kw in (dict(widen_coeff=0.5), dict(widen_coeff=4.0, mcts_batch_size=96), dict(n_games=0)):
        with pytest_raises(E.EngineError):
            mk(kw.pop("n_games", 1), **kw)
    with pytest_raises(E.EngineError):
        eng.reset([0], ["this is not a fen"], [None])
    with pytest_raises(E.EngineError):
        eng.reset([0], [None], ["e2e4 zz99"])
    with pytest_raises(E.EngineError):
        eng.reset([5], [None], [None])
    # a game longer than max_plies sets a status bit instead of writing out of bounds
    small = mk(1, num_simulations=8, mcts_batch_size=8, max_plies=6)
    small.reset([0])
    ss = Searcher(backend, small)
    for _ in range(8):
        nl, term, ply = small.root_info()
        res = ss.search([1], [T.softmax_eval(1)], [np.random.RandomState(0)], 0.1) if small.status()["status"][0] == 0 else None
        small.play(np.array([-2], dtype=np.int32))
    assert small.status()["status"][0] & 8 and small.root_info()[2][0] <= 5


class pytest_raises:
    def __init__(self, exc):
        self.exc = exc

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, self.exc), f"expected {self.exc}"
        return True


SEARCH_CONFIG_SWEEP = [
    dict(sims=7, batch=96),                                     # fewer simulations than one batch
    dict(sims=100, batch=1),                                    # one row per evaluation
    dict(sims=97, batch=13, cpuct=2.5, widen_coeff=1.0),        # ragged last batch, the narrowest widening the engine takes
    dict(sims=64, batch=8, widen_coeff=4.0),                    # widening faster than the child cap grows
    dict(sims=80, batch=16, dirichlet_alpha=0.0),               # no root noise
    dict(sims=80, batch=16, dirichlet_alpha=0.9, dirichlet_eps=1.0, cpuct=0.3),  # priors replaced by noise, exploitation-heavy
]


START_FENS = [
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R b KQkq - 0 1",   # black to move, all castling rights, 40+ moves
    "rnbqkbnr/ppp1pppp/8/8/3pP3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 3",           # en passant available at the root
    "8/P6k/8/8/8/8/p6K/8 w - - 0 60",                                          # promotions (4 pieces each) next move, move 60: low temperature
    "k7/8/1K6/8/8/8/8/7R w - - 96 80",                                         # fifty-move rule inside the search horizon
    "4k3/8/8/8/8/2n5/8/R3K2R w KQ - 3 30",                                     # in check at the root, temperature threshold move
    "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 1",                                   # back-rank mate in one for white
]


def check_games_from_positions_vs_oracle(backend, plies=5, sims=40, batch=8, scale=5.0):
    """Self-play games started from positions with the features the start position lacks: side to move, en passant,
    promotion, check, a ticking halfmove clock, forced mates (terminal simulations), fullmove numbers around the temperature
    threshold.  Moves, pi, dense states and z bit-exact against the oracle."""
    from fake_model import fake_logits_values

    def eval_fn_for(salt):
        def fn(planes):
            logits, v = fake_logits_values(planes, scale, salt)
            x = logits.astype(np.float64)
            e = np.exp(x - x.max(axis=1, keepdims=True))
            return (e / e.sum(axis=1, keepdims=True)).astype(np.float32), v
        return fn

    n = len(START_FENS)
    cfg = dict(num_simulations=sims, batch_size=batch, max_game_moves=plies)
    eng = make_engine(backend, n, cfg)
    eng.reset(list(range(n)), START_FENS)
    fns = [eval_fn_for(77 + g) for g in range(n)]
    got = play_games(backend, eng, fns, [np.random.RandomState(40 + g) for g in range(n)], 0.1, plies)
    ocfg = O.default_config(**cfg)
    for g in range(n):
        ref = O.self_play(fns[g], np.random.RandomState(40 + g), ocfg, start_fen=START_FENS[g])
        assert ref is not None, START_FENS[g]
        assert [O.move_to_uci(m) for m in ref["moves"]] == got[g]["moves"], START_FENS[g]
        for (st, pi, z), gpi, gst, gz in zip(ref["records"], got[g]["pis"], got[g]["states"], got[g]["z"]):
            assert np.array_equal(pi.view(np.uint32), gpi.view(np.uint32)), START_FENS[g]
            assert np.array_equal(st, gst), START_FENS[g]
            assert z == gz


# ---- long runs of terminal simulations (mcts.py:235-238) -----------------------------------------------------------
LONG_TERMINAL_RUN_CASES = [
    ("k7/8/1K6/8/8/8/8/7R w - - 0 1", [], 800, 96),                      # mate in one: hundreds of visits end in the same mated leaf
    ("7k/5Q2/6K1/8/8/8/8/8 w - - 0 1", [], 1500, 96),                      # several mating moves and stalemating ones side by side
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 98 80", [], 700, 64),                    # claimable fifty-move draws one ply below the root
    (O.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1".split(), 600, 96),  # claimable repetitions
]


def check_long_terminal_runs_vs_oracle(backend, case):
    """Whole trees, pi and best move against the oracle for searches dominated by terminal simulations (the register-resident
    burst path of the step kernel, including its table-window refills after 256 simulated visits)."""
    import test_fast_mode_emu as T

    fen, moves, sims, batch = case
    cfg = dict(num_simulations=sims, batch_size=batch)
    eng = make_engine(backend, 1, cfg)
    eng.reset([0], [fen], [" ".join(moves) or None])
    fn = T.softmax_eval(3, scale=4.0)
    res = Searcher(backend, eng).search([1], [fn], [np.random.RandomState(5)], 0.1)
    b = O.Board(fen)
    trk = O.PyTracker(); trk.add_board(b)
    for u in moves:
        b.push(u); trk.add_board(b)
    pos = b.positions()
    r = O.run_mcts(b, pos[max(0, len(pos) - 8):-1], trk, fn, np.random.RandomState(5), O.default_config(**cfg))
    assert r["n_terminal_sims"] >= sims // 2, r["n_terminal_sims"]   # the case does exercise long terminal runs
    assert E.move_to_uci(int(res["best_move"][0])) == O.move_to_uci(r["best"])
    assert np.array_equal(dense_pi(res, 0).view(np.uint32), r["pi"].view(np.uint32))
    exp = {"/".join(k): list(v) for k, v in O.canonical_tree(r["nodes"]).items()}
    assert canonical_tree(eng.debug_tree(0)) == exp
    assert int(eng.status()["term_sims"][0]) == r["n_terminal_sims"]


def check_watched_status_word(backend):
    """bo_engine_watch: an int32 device word named to the engine arrives with every fetched result block (no copy or wait of its
    own); bo_engine_watch_seen reports the OR of the values seen and clears on request.  This is how the self-play loop learns
    once per ply that the split-precision tower had to saturate an activation."""
    eng = make_engine(backend, 2, dict(num_simulations=20, batch_size=8, dirichlet_alpha=0.0))
    eng.reset([0, 1])
    sr = Searcher(backend, eng)
    ev = lambda planes: (np.full((1, E.NUM_ACTIONS), 1.0 / E.NUM_ACTIONS, np.float32), np.zeros(1, np.float32))
    word = Buf(backend, (4,))          # any 4-byte device word (its bits are what is watched)
    eng.watch(word.ptr)
    go = np.ones(2, np.int32)
    sr.search(go, [ev, ev], [None, None], 0.0)
    assert eng.watch_seen() == 0
    word.set(np.array([1, 0, 0, 0], np.int32).view(np.float32))
    sr.search(go, [ev, ev], [None, None], 0.0)
    assert eng.watch_seen(clear=False) == 1 and eng.watch_seen() == 1 and eng.watch_seen() == 0
    eng.watch(0)                          # unwatched again: the word's value no longer arrives
    sr.search(go, [ev, ev], [None, None], 0.0)
    assert eng.watch_seen() == 0
    eng.close()
