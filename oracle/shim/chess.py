"""
oracle/shim/chess.py -- TEST INFRASTRUCTURE (not the product, not a copy of python-chess).

A minimal module named ``chess`` exposing just the python-chess API surface that the reference's
hot path touches (SURVEY.md section 8c lists every call site), backed by the CPU oracle's own
rules engine (oracle/bo_rules.c).  It exists for two purposes only:

  1. tests/golden/generate_golden.py puts this directory and /root/reference on sys.path so the
     UNMODIFIED reference mcts.py / utils.py / self_play.py execute in this container and emit
     golden traces (python-chess itself is not installed and cannot be installed here);
  2. tests exercise the product's drop-in surface (betaone_amd.dropin.mcts.run_mcts(board, ...))
     with duck-typed boards, exactly as uci.py would with real python-chess boards.

What it pins and what it does not: traces produced through this shim pin the reference's tree
arithmetic / control flow / encodings GIVEN these rules; they do not pin the rules against
python-chess ("parity unpinned" items).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import oracle as _O  # noqa: E402

Color = bool
PieceType = int
Square = int
Bitboard = int

WHITE, BLACK = True, False
COLORS = [WHITE, BLACK]
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = 1, 2, 3, 4, 5, 6
PIECE_TYPES = [PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING]
PIECE_SYMBOLS = [None, "p", "n", "b", "r", "q", "k"]
FILE_NAMES = "abcdefgh"
RANK_NAMES = "12345678"
STARTING_FEN = _O.STARTING_FEN
SQUARES = list(range(64))
SQUARE_NAMES = [f + r for r in RANK_NAMES for f in FILE_NAMES]
(A1, B1, C1, D1, E1, F1, G1, H1) = range(8)
(A8, B8, C8, D8, E8, F8, G8, H8) = range(56, 64)


def square(file_index: int, rank_index: int) -> int:
    return rank_index * 8 + file_index


def square_file(sq: int) -> int:
    return sq & 7


def square_rank(sq: int) -> int:
    return sq >> 3


def square_name(sq: int) -> str:
    return SQUARE_NAMES[sq]


def parse_square(name: str) -> int:
    return SQUARE_NAMES.index(name)


class Piece:
    def __init__(self, piece_type: int, color: bool):
        self.piece_type, self.color = piece_type, color

    def symbol(self) -> str:
        s = PIECE_SYMBOLS[self.piece_type]
        return s.upper() if self.color else s

    def __eq__(self, o):
        return isinstance(o, Piece) and (self.piece_type, self.color) == (o.piece_type, o.color)

    def __hash__(self):
        return hash((self.piece_type, self.color))


class Move:
    def __init__(self, from_square: int, to_square: int, promotion=None, drop=None):
        self.from_square, self.to_square = from_square, to_square
        self.promotion = promotion if promotion else None
        self.drop = drop

    def uci(self) -> str:
        if not self:
            return "0000"
        s = SQUARE_NAMES[self.from_square] + SQUARE_NAMES[self.to_square]
        return s + (PIECE_SYMBOLS[self.promotion] if self.promotion else "")

    @classmethod
    def from_uci(cls, uci: str) -> "Move":
        if uci == "0000":
            return cls(0, 0)
        if len(uci) not in (4, 5):
            raise ValueError(f"expected uci string to be of length 4 or 5: {uci!r}")
        promo = PIECE_SYMBOLS.index(uci[4]) if len(uci) == 5 else None
        return cls(SQUARE_NAMES.index(uci[0:2]), SQUARE_NAMES.index(uci[2:4]), promotion=promo)

    def __bool__(self):
        return bool(self.from_square or self.to_square or self.promotion)

    def __eq__(self, o):
        return isinstance(o, Move) and (self.from_square, self.to_square, self.promotion) == (
            o.from_square, o.to_square, o.promotion)

    def __hash__(self):
        return hash((self.from_square, self.to_square, self.promotion))

    def __repr__(self):
        return f"Move.from_uci({self.uci()!r})"

    __str__ = uci


def _cmove(m: Move) -> _O.Move:
    c = _O.Move()
    c.from_sq, c.to_sq, c.promo = m.from_square, m.to_square, m.promotion or 0
    return c


class Board:
    def __init__(self, fen: str | None = STARTING_FEN, *, chess960: bool = False):
        self._L = _O.lib()
        self._s = _O.Stack()
        self.move_stack = []
        self.chess960 = False
        p = _O.Pos()
        if fen is None:
            fen = "8/8/8/8/8/8/8/8 w - - 0 1"
        if self._L.bo_pos_from_fen(fen.encode(), C.byref(p)) != 0:
            raise ValueError(f"invalid fen: {fen!r}")
        self._L.bo_stack_init(C.byref(self._s), C.byref(p))

    def __del__(self):
        try:
            self._L.bo_stack_free(C.byref(self._s))
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------
    @property
    def _p(self) -> _O.Pos:
        return self._s.pos[self._s.n - 1]

    turn = property(lambda self: bool(self._p.turn))
    halfmove_clock = property(lambda self: int(self._p.halfmove_clock))
    fullmove_number = property(lambda self: int(self._p.fullmove_number))
    ep_square = property(lambda self: None if self._p.ep_square < 0 else int(self._p.ep_square))
    castling_rights = property(lambda self: int(self._p.castling))
    pawns = property(lambda self: int(self._p.pawns))
    knights = property(lambda self: int(self._p.knights))
    bishops = property(lambda self: int(self._p.bishops))
    rooks = property(lambda self: int(self._p.rooks))
    queens = property(lambda self: int(self._p.queens))
    kings = property(lambda self: int(self._p.kings))
    occupied = property(lambda self: int(self._p.occ[0] | self._p.occ[1]))

    @property
    def occupied_co(self):
        return [int(self._p.occ[0]), int(self._p.occ[1])]

    def reset(self):
        p = _O.Pos()
        self._L.bo_pos_from_fen(STARTING_FEN.encode(), C.byref(p))
        self._L.bo_stack_free(C.byref(self._s))
        self._L.bo_stack_init(C.byref(self._s), C.byref(p))
        self.move_stack = []

    def set_fen(self, fen: str):
        p = _O.Pos()
        if self._L.bo_pos_from_fen(fen.encode(), C.byref(p)) != 0:
            raise ValueError(f"invalid fen: {fen!r}")
        self._L.bo_stack_free(C.byref(self._s))
        self._L.bo_stack_init(C.byref(self._s), C.byref(p))
        self.move_stack = []

    def copy(self, *, stack=True) -> "Board":
        b = Board.__new__(Board)
        b._L = self._L
        b._s = _O.Stack()
        b.chess960 = False
        if stack is True:
            self._L.bo_stack_copy(C.byref(b._s), C.byref(self._s))
            b.move_stack = list(self.move_stack)
        else:
            p = self._p.copy()
            self._L.bo_stack_init(C.byref(b._s), C.byref(p))
            b.move_stack = []
        return b

    __copy__ = copy

    def fen(self, *, en_passant: str = "legal", **_kw) -> str:
        buf = C.create_string_buffer(128)
        self._L.bo_pos_to_fen(C.byref(self._p), buf, 128)
        f = buf.value.decode()
        if en_passant == "fen" and self.ep_square is not None:  # raw ep square, capturable or not
            parts = f.split()
            parts[3] = SQUARE_NAMES[self.ep_square]
            f = " ".join(parts)
        return f

    def clean_castling_rights(self) -> int:
        return self.castling_rights

    def _transposition_key(self):
        return self._s.key[self._s.n - 1].tup()

    def __eq__(self, o):
        if not isinstance(o, Board):
            return NotImplemented
        return bytes(self._p) == bytes(o._p)

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    __hash__ = None

    # -- pieces --------------------------------------------------------------------------------
    def pieces_mask(self, piece_type: int, color: bool) -> int:
        bbs = [0, self.pawns, self.knights, self.bishops, self.rooks, self.queens, self.kings]
        return bbs[piece_type] & int(self._p.occ[1 if color else 0])

    def pieces(self, piece_type: int, color: bool):
        m = self.pieces_mask(piece_type, color)
        return [sq for sq in range(64) if (m >> sq) & 1]

    def piece_type_at(self, sq: int):
        t = self._L.bo_piece_type_at(C.byref(self._p), sq)
        return t or None

    def piece_at(self, sq: int):
        t = self._L.bo_piece_type_at(C.byref(self._p), sq)
        if not t:
            return None
        return Piece(t, bool(self._L.bo_color_at(C.byref(self._p), sq)))

    def has_kingside_castling_rights(self, color: bool) -> bool:
        return bool(self._p.castling & (1 << (7 if color else 63)))

    def has_queenside_castling_rights(self, color: bool) -> bool:
        return bool(self._p.castling & (1 << (0 if color else 56)))

    def has_legal_en_passant(self) -> bool:
        return bool(self._L.bo_has_legal_en_passant(C.byref(self._p)))

    def is_check(self) -> bool:
        return bool(self._L.bo_is_check(C.byref(self._p)))

    # -- moves ---------------------------------------------------------------------------------
    @property
    def legal_moves(self):
        arr = (_O.Move * _O.MAX_MOVES)()
        n = self._L.bo_legal_moves(C.byref(self._p), arr)
        return [Move(arr[i].from_sq, arr[i].to_sq, arr[i].promo or None) for i in range(n)]

    def is_legal(self, move: Move) -> bool:
        return move in self.legal_moves

    def push(self, move: Move):
        assert self._L.bo_piece_type_at(C.byref(self._p), move.from_square), \
            f"push() expects move to be pseudo-legal, but got {move} in {self.fen()}"
        self._L.bo_stack_push(C.byref(self._s), _cmove(move))
        self.move_stack.append(move)

    def pop(self) -> Move:
        self._L.bo_stack_pop(C.byref(self._s))
        return self.move_stack.pop()

    def parse_uci(self, uci: str) -> Move:
        m = Move.from_uci(uci)
        if m not in self.legal_moves:
            raise ValueError(f"illegal uci: {uci!r} in {self.fen()}")
        return m

    def push_uci(self, uci: str) -> Move:
        m = self.parse_uci(uci)
        self.push(m)
        return m

    # -- outcome -------------------------------------------------------------------------------
    def _termination(self, claim_draw: bool) -> int:
        t = self._L.bo_termination_claim_draw(C.byref(self._s))
        if not claim_draw and t in (6, 7):
            return 0
        return t

    def is_game_over(self, *, claim_draw: bool = False) -> bool:
        return self._termination(claim_draw) != 0

    def is_checkmate(self) -> bool:
        return self._termination(False) == 1

    def result(self, *, claim_draw: bool = False) -> str:
        t = self._termination(claim_draw)
        if t == 0:
            return "*"
        if t == 1:
            return "0-1" if self.turn == WHITE else "1-0"
        return "1/2-1/2"

    def __repr__(self):
        return f"Board({self.fen()!r})"


_O.lib().bo_stack_copy.argtypes = [C.POINTER(_O.Stack), C.POINTER(_O.Stack)]
_O.lib().bo_stack_copy.restype = None
