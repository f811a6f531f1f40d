"""CPU tests of the oracle's rules engine against public known-answers (G3 of SURVEY.md section 8c).
These pin the SET of legal moves; python-chess's move ORDER is restated, not pinned -- except for the start position, whose order the
package's documentation prints (test_startpos_order_is_python_chess_order); scripts/pin_python_chess.py compares everything with the real
package wherever it is installed."""
import pytest

from oracle import oracle as O

PERFT = [
    (O.STARTING_FEN, [20, 400, 8902, 197281, 4865609]),
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", [48, 2039, 97862, 4085603]),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", [14, 191, 2812, 43238, 674624]),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", [6, 264, 9467, 422333]),
    ("r2q1rk1/pP1p2pp/Q4n2/bbp1p3/Np6/1B3NBn/pPPP1PPP/R3K2R b KQ - 0 1", [6, 264, 9467, 422333]),
    ("rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", [44, 1486, 62379, 2103487]),
    ("r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", [46, 2079, 89890, 3894594]),
]


@pytest.mark.parametrize("fen,expected", PERFT)
def test_perft_known_answers(fen, expected):
    b = O.Board(fen)
    assert [b.perft(d + 1) for d in range(len(expected))] == expected


def test_startpos_order_is_python_chess_order():
    """A PUBLISHED vector of python-chess's move ORDER, the one layer the reference does not contain: the package's own documentation
    (README / docs "Core" quick start) prints the start position's generator as
        >>> board.legal_moves
        <LegalMoveGenerator at ... (Nh3, Nf3, Nc3, Na3, h3, g3, f3, e3, d3, c3, b3, a3, h4, g4, f4, e4, d4, c4, b4, a4)>
    i.e. non-pawn pieces from h8 down to a1 (targets high -> low), then single pawn pushes, then double pushes -- the list below in UCI.
    Everything else about the order is restated from the package's algorithm (oracle/bo_rules.c) and waits for
    scripts/pin_python_chess.py to be run where the package can be installed."""
    got = [O.move_to_uci(m) for m in O.Board().legal_moves()]
    assert got == ("g1h3 g1f3 b1c3 b1a3 h2h3 g2g3 f2f3 e2e3 d2d3 c2c3 b2b3 a2a3 "
                   "h2h4 g2g4 f2f4 e2e4 d2d4 c2c4 b2b4 a2a4").split()


def test_castling_after_pieces_before_pawns_and_promotion_order():
    b = O.Board("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1")
    mv = [O.move_to_uci(m) for m in b.legal_moves()]
    assert mv.index("e1g1") + 1 == mv.index("e1c1")          # h-side rook scanned first
    assert mv.index("a1b1") < mv.index("e1g1") < mv.index("d5e6")  # pieces, castling, pawn captures
    p = O.Board("rnbqkbnr/pppp1Ppp/8/8/8/8/PPPP1PPP/RNBQKBNR w KQkq - 0 1")
    mv = [O.move_to_uci(m) for m in p.legal_moves()]
    i = mv.index("f7g8q")
    assert mv[i:i + 4] == ["f7g8q", "f7g8r", "f7g8b", "f7g8n"]  # Q, R, B, N
    assert mv.index("f7g8q") < mv.index("f7e8q")                 # capture targets high -> low


def test_evasions_put_king_moves_first():
    b = O.Board("rnbqkbnr/ppp2ppp/8/1B1pp3/4P3/8/PPPP1PPP/RNBQK1NR b KQkq - 1 3")
    mv = [O.move_to_uci(m) for m in b.legal_moves()]
    assert mv[0] == "e8e7"
    assert set(mv) == {"e8e7", "d8d7", "c8d7", "b8d7", "b8c6", "c7c6"}


def test_transposition_key_ep_only_when_capturable():
    b = O.Board()
    b.push("e2e4")
    assert b.pos.ep_square == 20 and b.key().ep == -1       # raw ep square set, but no capture possible
    b.push("a7a6"); b.push("e4e5"); b.push("d7d5")
    assert b.pos.ep_square == 43 and b.key().ep == 43        # exd6 is legal
    assert b.fen().split()[3] == "d6"


TERMINATIONS = [
    ("k6R/8/1K6/8/8/8/8/8 b - - 1 1", [], 1),                       # checkmate
    ("8/8/8/8/8/2k5/8/K7 w - - 0 1", [], 2),                        # K vs K
    ("8/8/8/8/8/2k5/8/KB6 w - - 0 1", [], 2),                       # K+B vs K
    ("7k/5Q2/6K1/8/8/8/8/8 b - - 0 1", [], 3),                      # stalemate
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 150 120", [], 4),                 # 75-move
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 100 80", [], 6),                  # 50-move claim
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 99 80", [], 6),                   # claim after the next move
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 98 80", [], 0),
    (O.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1".split(), 7),   # Ng8 would repeat 3x
    (O.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6".split(), 0),
    (O.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1 f6g8".split(), 7),  # 3rd occurrence on board
    (O.STARTING_FEN, ("g1f3 g8f6 f3g1 f6g8 " * 4).split(), 5),           # fivefold
]


@pytest.mark.parametrize("fen,moves,term", TERMINATIONS)
def test_outcome_claim_draw(fen, moves, term):
    b = O.Board(fen)
    for m in moves:
        b.push(m)
    assert b.termination() == term, O.TERMINATIONS[b.termination()]


def test_irreversible_move_cuts_repetition_chain():
    # losing castling rights makes earlier positions unreachable for the repetition count
    b = O.Board("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
    for m in "e1e2 e8e7 e2e1 e7e8 e1e2 e8e7 e2e1 e7e8".split():
        b.push(m)
    # position after the first king round trip has NO castling rights: differs from the start
    assert b.termination() == 0
    for m in "e1e2 e8e7 e2e1".split():
        b.push(m)
    assert b.termination() == 7
