"""
oracle/oracle.py -- ctypes front-end of the CPU ORACLE (test infrastructure, NOT the product).

Wraps oracle/libbo_oracle.so (bo_rules.c / bo_codec.c / bo_mcts.c), the plain-C restatement of
the reference's self-play rollout path (mcts.py, self_play.py, utils.py and the python-chess
rules they call).  NumPy pieces of the reference that are not worth restating in C are mirrored
here with the *same NumPy calls* the reference makes:

  apply_temperature / select_move_with_temperature   /root/reference/self_play.py:25-80
  np.random.dirichlet([alpha]*n)                      /root/reference/mcts.py:192

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbo_oracle.so")

NUM_ACTIONS = 4672
INPUT_CHANNELS = 120
PLANES_SIZE = INPUT_CHANNELS * 64
MAX_MOVES = 256

PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = 1, 2, 3, 4, 5, 6
WHITE, BLACK = 1, 0

TERMINATIONS = {0: None, 1: "checkmate", 2: "insufficient_material", 3: "stalemate",
                4: "seventyfive_moves", 5: "fivefold_repetition", 6: "fifty_moves",
                7: "threefold_repetition"}


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc, strict binary32) if needed."""
    srcs = [os.path.join(_HERE, f) for f in ("bo_rules.c", "bo_codec.c", "bo_mcts.c", "bo_rules.h", "bo_oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libbo_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


class Pos(C.Structure):
    _fields_ = [("pawns", C.c_uint64), ("knights", C.c_uint64), ("bishops", C.c_uint64),
                ("rooks", C.c_uint64), ("queens", C.c_uint64), ("kings", C.c_uint64),
                ("occ", C.c_uint64 * 2), ("castling", C.c_uint64),
                ("turn", C.c_int32), ("ep_square", C.c_int32),
                ("halfmove_clock", C.c_int32), ("fullmove_number", C.c_int32)]

    def copy(self) -> "Pos":
        p = Pos()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Pos))
        return p


class Move(C.Structure):
    _fields_ = [("from_sq", C.c_uint8), ("to_sq", C.c_uint8), ("promo", C.c_uint8), ("pad", C.c_uint8)]

    def tup(self) -> Tuple[int, int, int]:
        return (self.from_sq, self.to_sq, self.promo)


class Key(C.Structure):
    _fields_ = [("pawns", C.c_uint64), ("knights", C.c_uint64), ("bishops", C.c_uint64),
                ("rooks", C.c_uint64), ("queens", C.c_uint64), ("kings", C.c_uint64),
                ("occ_w", C.c_uint64), ("occ_b", C.c_uint64), ("castling", C.c_uint64),
                ("turn", C.c_int32), ("ep", C.c_int32)]

    def tup(self):
        """Same layout as python-chess Board._transposition_key()."""
        return (self.pawns, self.knights, self.bishops, self.rooks, self.queens, self.kings,
                self.occ_w, self.occ_b, bool(self.turn), self.castling, self.ep if self.ep >= 0 else None)

    @staticmethod
    def from_tup(t) -> "Key":
        k = Key()
        (k.pawns, k.knights, k.bishops, k.rooks, k.queens, k.kings, k.occ_w, k.occ_b) = t[:8]
        k.turn = 1 if t[8] else 0
        k.castling = t[9]
        k.ep = -1 if t[10] is None else int(t[10])
        return k


class Stack(C.Structure):
    _fields_ = [("pos", C.POINTER(Pos)), ("key", C.POINTER(Key)), ("irrev_in", C.POINTER(C.c_uint8)),
                ("n", C.c_int), ("cap", C.c_int)]


class Tracker(C.Structure):
    _fields_ = [("keys", C.POINTER(Key)), ("counts", C.POINTER(C.c_int)), ("n", C.c_int), ("cap", C.c_int)]


class Config(C.Structure):
    _fields_ = [("num_simulations", C.c_int), ("batch_size", C.c_int), ("cpuct", C.c_double),
                ("widen_coeff", C.c_double), ("dirichlet_alpha", C.c_double),
                ("dirichlet_eps", C.c_double), ("max_game_moves", C.c_int)]


EVAL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))
NOISE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double))
CHOOSE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int)


class Callbacks(C.Structure):
    _fields_ = [("eval", EVAL_CB), ("noise", NOISE_CB), ("choose", CHOOSE_CB), ("user", C.c_void_p)]


class OracleNode(C.Structure):
    _fields_ = [("parent", C.c_int32), ("n_visits", C.c_int32), ("q_value", C.c_float), ("prior", C.c_float),
                ("move", Move), ("n_children", C.c_int32), ("terminal", C.c_int32)]


class SearchResult(C.Structure):
    _fields_ = [("status", C.c_int), ("best_move", Move), ("pi", C.c_float * NUM_ACTIONS),
                ("n_nodes", C.c_int), ("nodes", C.POINTER(OracleNode)), ("n_evals", C.c_int),
                ("n_batches", C.c_int), ("n_terminal_sims", C.c_int), ("n_batch_rows", C.c_int),
                ("max_unique_in_batch", C.c_int)]


class Record(C.Structure):
    _fields_ = [("state", C.c_float * PLANES_SIZE), ("pi", C.c_float * NUM_ACTIONS), ("z", C.c_float)]


class GameResult(C.Structure):
    _fields_ = [("status", C.c_int), ("n_records", C.c_int), ("records", C.POINTER(Record)),
                ("n_moves", C.c_int), ("moves", C.POINTER(Move)), ("outcome", C.c_float),
                ("termination", C.c_int), ("n_sims", C.c_long), ("n_evals", C.c_long)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.bo_rules_init.restype = None
    L.bo_pos_from_fen.argtypes = [C.c_char_p, C.POINTER(Pos)]
    L.bo_pos_from_fen.restype = C.c_int
    L.bo_pos_to_fen.argtypes = [C.POINTER(Pos), C.c_char_p, C.c_int]
    L.bo_pos_to_fen.restype = None
    L.bo_legal_moves.argtypes = [C.POINTER(Pos), C.POINTER(Move)]
    L.bo_legal_moves.restype = C.c_int
    L.bo_is_check.argtypes = [C.POINTER(Pos)]
    L.bo_has_legal_en_passant.argtypes = [C.POINTER(Pos)]
    L.bo_is_insufficient_material.argtypes = [C.POINTER(Pos)]
    L.bo_push.argtypes = [C.POINTER(Pos), Move]
    L.bo_push.restype = None
    L.bo_is_zeroing.argtypes = [C.POINTER(Pos), Move]
    L.bo_is_irreversible.argtypes = [C.POINTER(Pos), Move]
    L.bo_key_of.argtypes = [C.POINTER(Pos), C.POINTER(Key)]
    L.bo_key_of.restype = None
    L.bo_piece_type_at.argtypes = [C.POINTER(Pos), C.c_int]
    L.bo_color_at.argtypes = [C.POINTER(Pos), C.c_int]
    L.bo_perft.argtypes = [C.POINTER(Pos), C.c_int]
    L.bo_perft.restype = C.c_uint64
    L.bo_stack_init.argtypes = [C.POINTER(Stack), C.POINTER(Pos)]
    L.bo_stack_init.restype = None
    L.bo_stack_free.argtypes = [C.POINTER(Stack)]
    L.bo_stack_free.restype = None
    L.bo_stack_push.argtypes = [C.POINTER(Stack), Move]
    L.bo_stack_push.restype = None
    L.bo_stack_pop.argtypes = [C.POINTER(Stack)]
    L.bo_stack_pop.restype = None
    L.bo_outcome_claim_draw.argtypes = [C.POINTER(Stack)]
    L.bo_termination_claim_draw.argtypes = [C.POINTER(Stack)]
    L.bo_move_to_index.argtypes = [Move]
    L.bo_index_to_move.argtypes = [C.c_int, C.POINTER(Pos), C.POINTER(Move)]
    L.bo_encode_board.argtypes = [C.POINTER(Pos), C.c_int, C.POINTER(Tracker), C.POINTER(C.c_float)]
    L.bo_encode_board.restype = None
    L.bo_tracker_add.argtypes = [C.POINTER(Tracker), C.POINTER(Key)]
    L.bo_tracker_add.restype = None
    L.bo_tracker_free.argtypes = [C.POINTER(Tracker)]
    L.bo_tracker_free.restype = None
    L.bo_tracker_count.argtypes = [C.POINTER(Tracker), C.POINTER(Key)]
    L.bo_np_sum_f32.argtypes = [C.POINTER(C.c_float), C.c_long]
    L.bo_np_sum_f32.restype = C.c_float
    L.bo_oracle_run_mcts.argtypes = [C.POINTER(Config), C.POINTER(Callbacks), C.POINTER(Stack), C.POINTER(Pos),
                                     C.c_int, C.POINTER(Tracker), C.POINTER(SearchResult)]
    L.bo_oracle_result_free.argtypes = [C.POINTER(SearchResult)]
    L.bo_oracle_result_free.restype = None
    L.bo_oracle_self_play.argtypes = [C.POINTER(Config), C.POINTER(Callbacks), C.c_char_p, C.c_int,
                                      C.POINTER(GameResult)]
    L.bo_game_result_free.argtypes = [C.POINTER(GameResult)]
    L.bo_game_result_free.restype = None
    L.bo_rules_init()
    _lib = L
    return L


STARTING_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"


def move_from_uci(s: str) -> Move:
    m = Move()
    m.from_sq = (ord(s[1]) - 49) * 8 + (ord(s[0]) - 97)
    m.to_sq = (ord(s[3]) - 49) * 8 + (ord(s[2]) - 97)
    m.promo = {"n": KNIGHT, "b": BISHOP, "r": ROOK, "q": QUEEN}[s[4]] if len(s) > 4 else 0
    return m


def move_to_uci(m) -> str:
    f, t, p = (m.from_sq, m.to_sq, m.promo) if isinstance(m, Move) else m
    s = "abcdefgh"[f & 7] + str((f >> 3) + 1) + "abcdefgh"[t & 7] + str((t >> 3) + 1)
    return s + (" pnbrqk"[p] if p else "")


class Board:
    """A board with its move stack (what python-chess's Board is to the reference)."""

    def __init__(self, fen: str = STARTING_FEN):
        self._L = lib()
        p = Pos()
        if self._L.bo_pos_from_fen(fen.encode(), C.byref(p)) != 0:
            raise ValueError(f"bad fen: {fen!r}")
        self.stack = Stack()
        self._L.bo_stack_init(C.byref(self.stack), C.byref(p))
        self.moves: List[Tuple[int, int, int]] = []

    def __del__(self):
        try:
            self._L.bo_stack_free(C.byref(self.stack))
        except Exception:
            pass

    @property
    def pos(self) -> Pos:
        return self.stack.pos[self.stack.n - 1]

    def positions(self) -> List[Pos]:
        return [self.stack.pos[i] for i in range(self.stack.n)]

    def fen(self) -> str:
        buf = C.create_string_buffer(128)
        self._L.bo_pos_to_fen(C.byref(self.pos), buf, 128)
        return buf.value.decode()

    def legal_moves(self) -> List[Move]:
        arr = (Move * MAX_MOVES)()
        n = self._L.bo_legal_moves(C.byref(self.pos), arr)
        out = []
        for i in range(n):
            m = Move()
            m.from_sq, m.to_sq, m.promo = arr[i].from_sq, arr[i].to_sq, arr[i].promo
            out.append(m)
        return out

    def push(self, m):
        if isinstance(m, str):
            m = move_from_uci(m)
        self._L.bo_stack_push(C.byref(self.stack), m)
        self.moves.append(m.tup())

    def pop(self):
        self._L.bo_stack_pop(C.byref(self.stack))
        self.moves.pop()

    def key(self) -> Key:
        return self.stack.key[self.stack.n - 1]

    def termination(self) -> int:
        return self._L.bo_termination_claim_draw(C.byref(self.stack))

    def perft(self, depth: int) -> int:
        return int(self._L.bo_perft(C.byref(self.pos), depth))


def move_to_index(m) -> int:
    if not isinstance(m, Move):
        mm = Move()
        mm.from_sq, mm.to_sq, mm.promo = m
        m = mm
    idx = lib().bo_move_to_index(m)
    if idx < 0:
        raise ValueError("move_to_index")
    return idx


def index_to_move(index: int, pos: Pos) -> Tuple[int, int, int]:
    m = Move()
    rc = lib().bo_index_to_move(int(index), C.byref(pos), C.byref(m))
    if rc != 0:
        raise ValueError(f"index_to_move rc={rc}")
    return m.tup()


class PyTracker:
    """utils.RepetitionTracker over exact keys (reference utils.py:68-107)."""

    def __init__(self):
        self.t = Tracker()

    def __del__(self):
        try:
            lib().bo_tracker_free(C.byref(self.t))
        except Exception:
            pass

    def add_key(self, k: Key):
        lib().bo_tracker_add(C.byref(self.t), C.byref(k))

    def add_board(self, b: Board):
        self.add_key(b.key())

    def count(self, k: Key) -> int:
        return lib().bo_tracker_count(C.byref(self.t), C.byref(k))


def encode_board(hist: Sequence[Pos], trk: PyTracker) -> np.ndarray:
    arr = (Pos * len(hist))(*hist)
    out = np.zeros(PLANES_SIZE, dtype=np.float32)
    lib().bo_encode_board(arr, len(hist), C.byref(trk.t), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(INPUT_CHANNELS, 8, 8)


def np_sum_f32(a: np.ndarray) -> np.float32:
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.float32(lib().bo_np_sum_f32(a.ctypes.data_as(C.POINTER(C.c_float)), a.size))


# ---------------------------------------------------------------------------------------------
# self_play.py:25-80 mirrored with the same NumPy calls (rng = a numpy RandomState standing in for
# the process-global legacy RNG the reference uses).
# ---------------------------------------------------------------------------------------------
def apply_temperature(probs: np.ndarray, temperature: float, rng) -> np.ndarray:
    """self_play.py:25-56 restated (same NumPy calls, same order)."""
    if temperature == 0:
        res = np.zeros_like(probs)
        where_max = np.where(probs == np.max(probs))[0]
        if len(where_max) > 0:
            res[rng.choice(where_max)] = 1.0
        return res
    if abs(temperature - 1.0) < 1e-6:
        return probs
    with np.errstate(divide="ignore", invalid="ignore"):
        x = np.power(probs.astype(np.float64), 1.0 / temperature)
    x[~np.isfinite(x)] = 0.0
    sx = np.sum(x)
    if not sx > 1e-9:
        nz = np.where(probs > 1e-9)[0]
        if len(nz) > 0:
            u = np.zeros_like(probs, dtype=np.float32)
            u[nz] = 1.0 / len(nz)
            return u
        return probs.astype(np.float32)
    y = (x / sx).astype(np.float32)
    sy = np.sum(y)
    if abs(sy - 1.0) > 1e-6 and sy > 1e-9:
        y /= sy
    return y


def select_move_with_temperature(probs: np.ndarray, move_number: int, rng, threshold=30, t_init=1.0,
                                 t_final=0.1) -> int:
    """self_play.py:59-80 restated."""
    scaled = apply_temperature(probs, t_init if move_number < threshold else t_final, rng)
    try:
        tot = np.sum(scaled)
        if abs(tot - 1.0) > 1e-6:
            if tot > 1e-9:
                scaled /= tot
            else:
                return int(np.argmax(probs))
        return int(rng.choice(len(scaled), p=scaled))
    except ValueError:
        return int(np.argmax(probs))


# ---------------------------------------------------------------------------------------------
# search / self-play drivers
# ---------------------------------------------------------------------------------------------
EvalFn = Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]]  # planes[n,120,8,8] -> probs[n,4672], values[n]


def default_config(**kw) -> Config:
    c = Config()
    c.num_simulations = kw.get("num_simulations", 250)
    c.batch_size = kw.get("batch_size", 96)
    c.cpuct = kw.get("cpuct", 1.0)
    c.widen_coeff = kw.get("widen_coeff", 1.5)
    c.dirichlet_alpha = kw.get("dirichlet_alpha", 0.1)
    c.dirichlet_eps = kw.get("dirichlet_eps", 0.25)
    c.max_game_moves = kw.get("max_game_moves", 16384)
    return c


class _CbHolder:
    def __init__(self, eval_fn: EvalFn, rng, cfg: Config, temperature=(30, 1.0, 0.1)):
        self.eval_fn, self.rng, self.cfg, self.temperature = eval_fn, rng, cfg, temperature
        self.error: Optional[BaseException] = None
        self.eval_log: List[int] = []

        def _eval(user, planes, n, probs, values):
            try:
                x = np.ctypeslib.as_array(planes, shape=(n, INPUT_CHANNELS, 8, 8))
                p, v = self.eval_fn(x)
                self.eval_log.append(n)
                np.ctypeslib.as_array(probs, shape=(n, NUM_ACTIONS))[:] = np.asarray(p, dtype=np.float32)
                np.ctypeslib.as_array(values, shape=(n,))[:] = np.asarray(v, dtype=np.float32).reshape(n)
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through C
                self.error = e
                return -10

        def _noise(user, n_legal, out):
            try:
                noise = self.rng.dirichlet([self.cfg.dirichlet_alpha] * n_legal)  # mcts.py:192
                np.ctypeslib.as_array(out, shape=(n_legal,))[:] = noise
                return 0
            except BaseException as e:  # noqa: BLE001
                self.error = e
                return -11

        def _choose(user, pi, fullmove_number):
            try:
                p = np.ctypeslib.as_array(pi, shape=(NUM_ACTIONS,)).copy()
                th, ti, tf = self.temperature
                return select_move_with_temperature(p, fullmove_number, self.rng, th, ti, tf)
            except BaseException as e:  # noqa: BLE001
                self.error = e
                return 0

        self.cbs = Callbacks(EVAL_CB(_eval), NOISE_CB(_noise), CHOOSE_CB(_choose), None)


def _tree_from_result(res: SearchResult):
    nodes = []
    for i in range(res.n_nodes):
        nd = res.nodes[i]
        nodes.append(dict(parent=nd.parent, n=nd.n_visits, q=np.float32(nd.q_value), prior=np.float32(nd.prior),
                          move=nd.move.tup(), n_children=nd.n_children, terminal=nd.terminal))
    return nodes


def canonical_tree(nodes) -> dict:
    """Order-independent form: {move-path tuple: (n, q bits, prior bits, n_children)}; root q excluded
    (the reference keeps root.q as a Python float that nothing reads)."""
    paths = {}
    out = {}
    for i, nd in enumerate(nodes):
        path = () if nd["parent"] < 0 else paths[nd["parent"]] + (move_to_uci(nd["move"]),)
        paths[i] = path
        qbits = None if i == 0 else int(np.float32(nd["q"]).view(np.uint32))
        out[path] = (int(nd["n"]), qbits, int(np.float32(nd["prior"]).view(np.uint32)), int(nd["n_children"]))
    return out


def run_mcts(board: Board, hist: Sequence[Pos], trk: PyTracker, eval_fn: EvalFn, rng, cfg: Optional[Config] = None):
    """mcts.run_mcts.  hist = positions BEFORE the root.  Returns dict(best, pi, nodes, stats)."""
    cfg = cfg or default_config()
    h = _CbHolder(eval_fn, rng, cfg)
    res = SearchResult()
    harr = (Pos * max(1, len(hist)))(*hist)
    rc = lib().bo_oracle_run_mcts(C.byref(cfg), C.byref(h.cbs), C.byref(board.stack), harr, len(hist),
                                  C.byref(trk.t), C.byref(res))
    if h.error is not None:
        raise h.error
    if rc != 0:
        raise RuntimeError(f"oracle run_mcts rc={rc}")
    try:
        if res.status == 1:
            raise ValueError("max() arg is an empty sequence")  # mcts.py:279 on a root without legal moves
        out = dict(best=res.best_move.tup(), pi=np.array(res.pi, dtype=np.float32), nodes=_tree_from_result(res),
                   n_evals=res.n_evals, n_batches=res.n_batches, n_terminal_sims=res.n_terminal_sims,
                   n_batch_rows=res.n_batch_rows, max_unique_in_batch=res.max_unique_in_batch)
    finally:
        lib().bo_oracle_result_free(C.byref(res))
    return out


def self_play(eval_fn: EvalFn, rng, cfg: Optional[Config] = None, start_fen: str = "", max_plies: int = 0,
              temperature=(30, 1.0, 0.1)):
    """self_play.run_self_play_game.  Returns dict(records=[(state, pi, z)], moves, outcome, ...) or None."""
    cfg = cfg or default_config()
    h = _CbHolder(eval_fn, rng, cfg, temperature)
    res = GameResult()
    rc = lib().bo_oracle_self_play(C.byref(cfg), C.byref(h.cbs), start_fen.encode(), max_plies, C.byref(res))
    if h.error is not None:
        raise h.error
    if rc != 0:
        return None
    try:
        recs = []
        for i in range(res.n_records):
            r = res.records[i]
            recs.append((np.array(r.state, dtype=np.float32).reshape(INPUT_CHANNELS, 8, 8),
                         np.array(r.pi, dtype=np.float32), float(r.z)))
        out = dict(records=recs, moves=[res.moves[i].tup() for i in range(res.n_moves)], outcome=float(res.outcome),
                   termination=res.termination, n_sims=int(res.n_sims), n_evals=int(res.n_evals))
    finally:
        lib().bo_game_result_free(C.byref(res))
    return out
