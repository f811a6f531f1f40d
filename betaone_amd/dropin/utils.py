"""
utils -- host-side mirror of /root/reference/utils.py for CALLERS of the rollout path (train.py's
datasets, uci.py's tracker bookkeeping).  The rollout path itself never calls these: its boards,
move codec, repetition tracking and plane encoding live on the GPU (betaone_amd/csrc/bo_chess.h,
bo_tree.h) and are checked against the reference through the golden traces.

Boards are duck-typed python-chess boards (train.py / uci.py import python-chess themselves):
only `pieces_mask`/bitboard attributes, `turn`, castling queries, clocks, `ep_square`,
`_transposition_key()` and `legal_moves` are touched.
"""
from collections import Counter
from typing import List

import numpy as np
import torch

import config

PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)
WHITE, BLACK = True, False
# utils.py:15-28
PIECE_ORDER = [(pt, c) for pt in (PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING) for c in (WHITE, BLACK)]
HISTORY_BLOCK_SIZE = len(PIECE_ORDER) + 2
# utils.py:34-62, as (d_rank, d_file) / (d_file, d_rank)
QUEEN_DIRECTIONS = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]
KNIGHT_DIRECTIONS = [(2, 1), (1, 2), (-1, 2), (-2, 1), (-2, -1), (-1, -2), (1, -2), (2, -1)]
PROMOTION_DIRECTIONS = [(-1, 1), (0, 1), (1, 1)]
PROMOTION_PIECES = [KNIGHT, BISHOP, ROOK]
ACTIONS_PLANES = 56 + 8 + 9

_FILES = "abcdefgh"


def _uci(from_sq: int, to_sq: int, promo) -> str:
    s = _FILES[from_sq & 7] + str((from_sq >> 3) + 1) + _FILES[to_sq & 7] + str((to_sq >> 3) + 1)
    return s + (" pnbrqk"[promo] if promo else "")


class RepetitionTracker:
    """utils.py:68-107: a Counter over python-chess transposition keys of the REAL game positions."""

    def __init__(self):
        self.counts = Counter()

    def add_board(self, board):
        self.counts[board._transposition_key()] += 1

    def remove_board(self, board):
        key = board._transposition_key()
        if self.counts[key] > 0:
            self.counts[key] -= 1
            if self.counts[key] == 0:
                del self.counts[key]

    def repetitions(self, board) -> int:
        return max(0, self.counts[board._transposition_key()] - 1)

    def get_count(self, board) -> int:
        return self.counts[board._transposition_key()]

    def reset(self):
        self.counts.clear()


def _piece_masks(board):
    """12 bitboards in PIECE_ORDER."""
    if hasattr(board, "pieces_mask"):
        return [int(board.pieces_mask(pt, c)) for pt, c in PIECE_ORDER]
    out = []
    for pt, c in PIECE_ORDER:
        m = 0
        for sq in board.pieces(pt, c):
            m |= 1 << sq
        out.append(m)
    return out


def _plane(mask: int) -> np.ndarray:
    bits = np.unpackbits(np.array([mask], dtype="<u8").view(np.uint8), bitorder="little")
    return bits.astype(np.float32).reshape(8, 8)


def encode_board(board, history: List, tracker: RepetitionTracker) -> torch.Tensor:
    """utils.py:111-217 (absolute orientation, oldest history block first, current board last)."""
    if not history or board != history[-1]:
        history = (list(history[:-1]) if history else []) + [board]  # utils.py:128-138 repair
    history = history[-8:]
    enc = np.zeros((config.INPUT_CHANNELS, 8, 8), dtype=np.float32)
    start = (8 - len(history)) * HISTORY_BLOCK_SIZE
    for i, hb in enumerate(history):
        base = start + i * HISTORY_BLOCK_SIZE
        for j, mask in enumerate(_piece_masks(hb)):
            if mask:
                enc[base + j] = _plane(mask)
        rep = tracker.repetitions(hb)
        enc[base + 12] = 1.0 if rep >= 1 else 0.0
        enc[base + 13] = 1.0 if rep >= 2 else 0.0
    enc[112] = 1.0 if board.turn == WHITE else 0.0
    enc[113] = 1.0 if board.has_kingside_castling_rights(WHITE) else 0.0
    enc[114] = 1.0 if board.has_queenside_castling_rights(WHITE) else 0.0
    enc[115] = 1.0 if board.has_kingside_castling_rights(BLACK) else 0.0
    enc[116] = 1.0 if board.has_queenside_castling_rights(BLACK) else 0.0
    enc[117] = float(board.halfmove_clock)
    enc[118] = float(board.fullmove_number)
    if board.ep_square is not None:
        enc[119, board.ep_square >> 3, board.ep_square & 7] = 1.0
    return torch.from_numpy(enc)


def move_to_index(move) -> int:
    """utils.py:221-281: from_square*73 + plane."""
    f, t, promo = move.from_square, move.to_square, move.promotion
    fr, ff = f >> 3, f & 7
    dr, df = (t >> 3) - fr, (t & 7) - ff
    if promo and promo != QUEEN:
        if fr == 6:
            key = (df, dr)
        elif fr == 1:
            key = (df, -dr)
        else:
            raise ValueError(f"Invalid underpromotion move: {_uci(f, t, promo)}")
        if key not in PROMOTION_DIRECTIONS or promo not in PROMOTION_PIECES:
            raise ValueError(f"Invalid underpromotion move: {_uci(f, t, promo)}")
        return f * ACTIONS_PLANES + 64 + PROMOTION_PIECES.index(promo) * 3 + PROMOTION_DIRECTIONS.index(key)
    if (abs(dr), abs(df)) in ((1, 2), (2, 1)):
        return f * ACTIONS_PLANES + 56 + KNIGHT_DIRECTIONS.index((dr, df))
    if abs(dr) == abs(df) or dr == 0 or df == 0:
        dist = max(abs(dr), abs(df))
        if dist == 0 or dist > 7:
            raise ValueError(f"Invalid queen/sliding move: {_uci(f, t, promo)} with delta {(dr, df)}")
        step = ((dr > 0) - (dr < 0), (df > 0) - (df < 0))
        return f * ACTIONS_PLANES + QUEEN_DIRECTIONS.index(step) * 7 + dist - 1
    raise ValueError(f"Unhandled move type for move: {_uci(f, t, promo)}")


def _make_move(board, from_sq: int, to_sq: int, promotion=None):
    """Build a move object of the caller's own chess library (python-chess when board came from it)."""
    import sys

    mod = sys.modules.get(type(board).__module__)
    move_cls = getattr(mod, "Move", None)
    if move_cls is None:
        import chess  # the caller's python-chess

        move_cls = chess.Move
    return move_cls(from_sq, to_sq, promotion=promotion)


def index_to_move(index: int, board):
    """utils.py:284-365 (no legality check; ValueError where the reference raises it)."""
    if not (0 <= index < config.NUM_ACTIONS):
        raise ValueError(f"Index {index} out of valid range [0, {config.NUM_ACTIONS - 1}]")
    f, plane = divmod(int(index), ACTIONS_PLANES)
    fr, ff = f >> 3, f & 7
    piece = board.piece_at(f)
    promo = None
    if plane < 56:
        dr, df = QUEEN_DIRECTIONS[plane // 7]
        dist = plane % 7 + 1
        tr, tf = fr + dr * dist, ff + df * dist
        if piece and piece.piece_type == PAWN and ((piece.color == WHITE and fr == 6 and tr == 7) or
                                                   (piece.color == BLACK and fr == 1 and tr == 0)):
            promo = QUEEN
        what = "queen move"
    elif plane < 64:
        dr, df = KNIGHT_DIRECTIONS[plane - 56]
        tr, tf = fr + dr, ff + df
        what = "knight move"
    else:
        off = plane - 64
        if not piece or piece.piece_type != PAWN:
            raise ValueError(f"Index {index} implies underpromotion but no pawn at {_FILES[ff]}{fr + 1}")
        df, dr_rel = PROMOTION_DIRECTIONS[off % 3]
        if piece.color == WHITE and fr == 6:
            dr = dr_rel
        elif piece.color == BLACK and fr == 1:
            dr = -dr_rel
        else:
            raise ValueError(f"Index {index} implies underpromotion from invalid rank {fr} for color {piece.color}")
        tr, tf = fr + dr, ff + df
        promo = PROMOTION_PIECES[off // 3]
        what = "underpromotion"
    if not (0 <= tr <= 7 and 0 <= tf <= 7):
        raise ValueError(f"Index {index} {what} decodes to off-board square ({tr}, {tf})")
    return _make_move(board, f, tr * 8 + tf, promo)


def get_legal_mask(board) -> torch.Tensor:
    """utils.py:368-382."""
    mask = torch.zeros(config.NUM_ACTIONS, dtype=torch.bool)
    for move in board.legal_moves:
        try:
            mask[move_to_index(move)] = True
        except ValueError as e:
            print(f"Warning: Could not get index for legal move {move.uci()}: {e}")
    return mask


def get_game_outcome(board):
    """utils.py:385-396: None while the game is running, else the result from the perspective of the
    player who just moved (+1.0 mate delivered, 0.0 any draw)."""
    if not board.is_game_over(claim_draw=True):
        return None
    result = board.result(claim_draw=True)
    mover_is_white = not board.turn
    if result == "1-0":
        return 1.0 if mover_is_white else -1.0
    if result == "0-1":
        return 1.0 if not mover_is_white else -1.0
    return 0.0


def test_move_indexing(board) -> int:
    """utils.py:399-464: self-consistency of the move codec on one position; returns the error count."""
    errors, seen = 0, {}
    for move in board.legal_moves:
        try:
            idx = move_to_index(move)
        except Exception:
            errors += 1
            continue
        if not (0 <= idx < config.NUM_ACTIONS) or idx in seen.values():
            errors += 1
        seen[move] = idx
    legal = list(board.legal_moves)
    for move, idx in seen.items():
        try:
            back = index_to_move(idx, board)
        except Exception:
            errors += 1
            continue
        if back != move or back not in legal:
            errors += 1
    mask_idx = set(np.where(get_legal_mask(board).numpy())[0].tolist())
    errors += len(mask_idx ^ set(seen.values()))
    return errors
