"""
betaone_amd/engine.py -- thin ctypes binding of the C ABI in include/betaone_engine.h.

The product only ever loads csrc/libbetaone_hip.so (hipcc, gfx950) and raises if it is missing or
cannot be loaded -- there is no CPU fallback and no backend switch in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
# PyTorch-ROCm bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so).  It must be loaded BEFORE
# libbetaone_hip.so so that the engine binds to that same runtime: two HIP runtimes in one process do not
# share a device context (the second one reports "no ROCm-capable device"), and the engine exchanges raw
# device pointers and stream handles with torch.
import torch  # noqa: F401

NUM_ACTIONS = 4672
INPUT_CHANNELS = 120
ROW_FLOATS = 120 * 64
MAX_LEGAL = 256
RES_CAP = 256
POLICY_NONE, POLICY_LOGITS, POLICY_PROBS = 0, 1, 2

STATUS_BITS = {1: "node overflow", 2: "depth overflow", 4: "NaN PUCT score", 8: "ply overflow",
               16: "illegal action", 32: "leaf cache overflow", 64: "tracker overflow"}
ST_PLY_OVERFLOW, ST_ILLEGAL_ACTION = 8, 16   # per-GAME conditions (the reference aborts that game only); the rest are engine faults
ST_NODE_OVERFLOW = 1  # reference mode: a fault (the node arrays are sized for the search).  Fast mode: the game's arena was full at an
#                       expansion or a re-root -- the leaf stayed unexpanded (its value was still backed up) / the subtree that did not
#                       fit was dropped; the search is valid, only narrower.  Sticky per slot until the slot is reset.


LAZY_BEGIN = "lazy-begin"  # selfplay_turn(lazy_begin=True): the begun searches' root info comes from selfplay_begun()


class EngineError(RuntimeError):
    pass


class BoConfig(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("num_simulations", C.c_int32), ("mcts_batch_size", C.c_int32),
                ("max_plies", C.c_int32), ("cpuct", C.c_double), ("widen_coeff", C.c_double),
                ("dirichlet_alpha", C.c_double), ("dirichlet_epsilon", C.c_double), ("mode", C.c_int32),
                ("leaves_per_step", C.c_int32), ("fast_arena_granules", C.c_int32)]


class BoPosition(C.Structure):
    _fields_ = [("bb", C.c_uint64 * 8), ("turn", C.c_int32), ("castling", C.c_uint32), ("ep_square", C.c_int32),
                ("ep_key", C.c_int32), ("halfmove_clock", C.c_int32), ("fullmove_number", C.c_int32)]


class BoB1LayerDesc(C.Structure):
    _fields_ = [("weights_dev", C.c_void_p), ("bias_dev", C.c_void_p), ("se_w1_dev", C.c_void_p), ("se_w2_dev", C.c_void_p),
                ("c_in", C.c_int32), ("c_in_x", C.c_int32), ("mode", C.c_int32), ("se_hidden", C.c_int32),
                ("weights_split_dev", C.c_void_p), ("inv_scale", C.c_float), ("reserved", C.c_int32)]


class BoNode(C.Structure):
    _fields_ = [("parent", C.c_int32), ("n_visits", C.c_int32), ("first_child", C.c_int32), ("n_children", C.c_int32),
                ("q_value", C.c_float), ("prior", C.c_float), ("move", C.c_int32), ("terminal", C.c_int32)]


_I32P = C.POINTER(C.c_int32)
_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)

_SYMBOLS = {  # include/betaone_engine.h: the drop-in boundary
    "bo_abi_version": (C.c_int, []),
    "bo_last_error": (C.c_char_p, []),
    "bo_engine_create": (C.c_int, [C.POINTER(BoConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "bo_engine_destroy": (None, [C.c_void_p]),
    "bo_games_reset": (C.c_int, [C.c_void_p, C.c_int, _I32P, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p]),
    "bo_games_reset_ex": (C.c_int, [C.c_void_p, C.c_int, _I32P, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                    C.POINTER(BoPosition), _I32P, C.POINTER(BoPosition), _I32P, _I32P, C.c_void_p]),
    "bo_root_info": (C.c_int, [C.c_void_p, _I32P, _I32P, _I32P, C.c_void_p]),
    "bo_search_begin": (C.c_int, [C.c_void_p, _I32P, _F64P, C.c_void_p, C.c_void_p]),
    "bo_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bo_step_heads": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p]),
    "bo_search_poll": (C.c_int, [C.c_void_p, _I32P, _I32P, _I32P, C.c_void_p]),
    "bo_search_stop": (C.c_int, [C.c_void_p, _I32P, _I32P, C.c_void_p]),
    "bo_search_result": (C.c_int, [C.c_void_p, _I32P, _I32P, _F32P, _I32P, _I32P, _I32P, C.c_void_p]),
    "bo_play": (C.c_int, [C.c_void_p, _I32P, C.c_void_p]),
    "bo_game_export": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(BoPosition), _I32P, C.c_int32, _I32P, C.c_void_p]),
    "bo_game_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bo_rng_seed": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    "bo_rng_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint32), _I32P, _I32P, _F64P]),
    "bo_selfplay_sample": (C.c_int, [C.c_void_p, _I32P, _I32P, C.c_int32, C.c_double, C.c_double, _I32P, _I32P, _F32P,
                                     _I32P, _I32P, C.c_void_p]),
    "bo_selfplay_turn": (C.c_int, [C.c_void_p, _I32P, _I32P, C.c_int32, C.c_double, C.c_double, _I32P, _I32P, _F32P, _I32P, _I32P, _I32P,
                                   C.c_void_p, _I32P, _I32P, _I32P, C.c_int32, _I32P, C.c_void_p]),
    "bo_selfplay_noise": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bo_selfplay_autoturn": (C.c_int, [C.c_void_p, _I32P, _I32P, C.c_int32, C.c_double, C.c_double, _I32P, C.c_void_p, C.c_int32, C.c_void_p]),
    "bo_selfplay_autoturn_ready": (C.c_int, [C.c_void_p, _I32P]),
    "bo_selfplay_autoturn_collect": (C.c_int, [C.c_void_p, _I32P, _I32P, _F32P, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P]),
    "bo_search_result_prefetch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bo_selfplay_begun": (C.c_int, [C.c_void_p, _I32P, _I32P, _I32P]),
    "bo_selfplay_begin": (C.c_int, [C.c_void_p, _I32P, C.c_void_p, _I32P, _I32P, _I32P, C.c_void_p]),
    "bo_records_encode": (C.c_int, [C.c_int, C.POINTER(BoPosition), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bo_fast_options": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "bo_nn_tower_status": (C.c_int, [C.c_void_p, _I32P, C.c_void_p]),
    "bo_nn_tower_word": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bo_replay_create": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bo_replay_add_game": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(BoPosition), C.c_int32, _I32P, _I32P, _F32P, _F32P, C.POINTER(C.c_int64), C.c_void_p]),
    "bo_replay_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bo_replay_sample": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bo_replay_destroy": (None, [C.c_void_p]),
    "bo_nn_b1_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bo_nn_b1_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bo_nn_b1_status": (C.c_int, [C.c_void_p, _I32P, C.c_void_p]),
    "bo_nn_b1_destroy": (None, [C.c_void_p]),
    "bo_engine_watch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bo_engine_watch_words": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "bo_nn_b1_word": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bo_engine_watch_seen": (C.c_int, [C.c_void_p, _I32P, C.c_int32]),
    "bo_engine_status": (C.c_int, [C.c_void_p, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P, C.c_void_p]),
    "bo_movegen_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(BoPosition), _I32P, _I32P, _I32P, C.c_void_p]),
    "bo_nn_bias_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bo_nn_se_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p]),
    "bo_nn_se_residual_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    "bo_nn_heads": (C.c_int, [C.c_void_p] * 11 + [C.c_int, C.c_int, C.c_void_p]),
    "bo_nn_heads_f16": (C.c_int, [C.c_void_p] * 11 + [C.c_int, C.c_int, C.c_void_p]),
    "bo_nn_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
    "bo_nn_conv3x3_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "bo_nn_tower_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.POINTER(C.c_void_p)]),
    "bo_nn_tower_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bo_device_wall_clock_khz": (C.c_int, [C.c_int, _I32P]),
    "bo_stream_create_cu_mask": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_void_p)]),
    "bo_stream_destroy": (C.c_int, [C.c_void_p]),
    "bo_nn_value_tail": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bo_nn_tower_destroy": (None, [C.c_void_p]),
}
# include/betaone_lab.h: introspection for the parity tests and in-kernel timing for bench.py / scripts/ (same library, not the boundary)
_LAB_SYMBOLS = {
    "bo_debug_tree": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(BoNode), C.c_int32, _I32P, C.c_void_p]),
    "bo_debug_fast": (C.c_int, [C.c_void_p, C.c_int, _I32P, C.c_int32, _I32P, C.c_void_p]),
    "bo_event_pair_overhead": (C.c_int, [_F64P, C.c_int32, C.c_void_p]),
    "bo_nn_b1_profile": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]),
    "bo_fast_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _I32P, C.c_int32, _F64P, C.POINTER(C.c_int64),
                               C.c_void_p]),
    "bo_debug_profile": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
    "bo_debug_stamp": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "bo_nn_tower_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bo_select_wide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
}

HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libbetaone_hip.so")


def bind(cdll: C.CDLL) -> C.CDLL:
    """Attach argument/return types for every symbol include/betaone_engine.h and include/betaone_lab.h declare."""
    for name, (res, args) in list(_SYMBOLS.items()) + list(_LAB_SYMBOLS.items()):
        fn = getattr(cdll, name)  # AttributeError here == library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return cdll


_hip_lib: Optional[C.CDLL] = None
ABI_VERSION = 6   # BO_ABI_VERSION of include/betaone_engine.h this binding was written against (tests/test_abi.py compares)
PROF_SLOTS = 16   # BO_PROF_SLOTS


def load_hip_library() -> C.CDLL:
    """The product's only loader: csrc/libbetaone_hip.so or an exception."""
    global _hip_lib
    if _hip_lib is None:
        from . import build as _build

        if not os.path.exists(HIP_LIB_PATH):
            raise EngineError(f"{HIP_LIB_PATH} is missing: build it with `python -m betaone_amd.build` "
                              "(hipcc --offload-arch=gfx950); betaone_amd has no CPU fallback")
        if _build.needs_build():
            raise EngineError(f"{HIP_LIB_PATH} was not built from the sources next to it (csrc/*.h changed since): "
                              "run `python -m betaone_amd.build`")
        _hip_lib = bind(C.CDLL(HIP_LIB_PATH))
        if _hip_lib.bo_abi_version() != ABI_VERSION:
            raise EngineError(f"libbetaone_hip.so ABI version {_hip_lib.bo_abi_version()} != {ABI_VERSION} (include/betaone_engine.h)")
    return _hip_lib


def runtime_device(requested) -> torch.device:
    """The torch device a product path runs on: an MI355X ('cuda' / 'cuda:N', index resolved once against torch's
    current device so that the engine, its NN rows and the model agree on every rank) or an exception."""
    dev = torch.device(requested)
    if dev.type != "cuda":
        raise EngineError(f"betaone_amd runs on an MI355X (device 'cuda' / 'cuda:N', got {requested!r}); there is no CPU path")
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


def cu_partition_masks(n_cu: int, parts: int, layout: str = "contiguous") -> list:
    """`parts` disjoint CU sets covering CUs [0, n_cu - n_cu % parts) as uint32 mask words (bit i = CU i): part p owns CUs
    [p * n_cu / parts, (p + 1) * n_cu / parts) ("contiguous") or the CUs i with i % parts == p ("interleaved")."""
    if parts < 1 or n_cu < parts:
        raise ValueError(f"cannot split {n_cu} CUs into {parts} parts")
    per, words = n_cu // parts, (n_cu + 31) // 32
    out = []
    for p in range(parts):
        m = np.zeros(words, dtype=np.uint32)
        idx = (np.arange(p * per, (p + 1) * per) if layout == "contiguous" else np.arange(n_cu) if layout == "full"  # ("full": a lab
               else np.arange(per) * parts + p)                            # layout -- every part all CUs, only the stream's own queue)
        np.bitwise_or.at(m, idx // 32, (np.uint32(1) << (idx % 32).astype(np.uint32)))
        out.append(m)
    return out


class MaskedStream:
    """A HIP stream confined to a CU set (bo_stream_create_cu_mask) with its torch view (`.stream`, a torch.cuda.ExternalStream)."""

    def __init__(self, device: torch.device, mask_words: np.ndarray):
        self.lib = load_hip_library()
        self.mask = np.ascontiguousarray(mask_words, dtype=np.uint32)
        h = C.c_void_p()
        rc = self.lib.bo_stream_create_cu_mask(int(device.index or 0), self.mask.ctypes.data_as(C.POINTER(C.c_uint32)), len(self.mask), C.byref(h))
        if rc != 0:
            raise EngineError(f"bo_stream_create_cu_mask: {self.lib.bo_last_error().decode()}")
        self.handle = h.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=device)

    def close(self):
        if self.handle:
            self.stream.synchronize()
            self.lib.bo_stream_destroy(C.c_void_p(self.handle))
            self.handle = None


def move_to_uci(m: int) -> str:
    f, t, p = m & 63, (m >> 6) & 63, (m >> 12) & 7
    s = "abcdefgh"[f & 7] + str((f >> 3) + 1) + "abcdefgh"[t & 7] + str((t >> 3) + 1)
    return s + (" pnbrqk"[p] if p else "")


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: np.ndarray, typ=_I32P):
    return a.ctypes.data_as(typ)


class PositionList:
    """positions[0..n) of an exported game: indexable like a list of BoPosition, backed by ONE ctypes array (`raw`)."""

    def __init__(self, raw, n: int):
        self.raw, self.n = raw, n

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self.raw[k] for k in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return self.raw[i]

    def __iter__(self):
        return (self.raw[k] for k in range(self.n))


class Engine:
    """G game slots on one GPU.  All pointer arguments are raw addresses (tensor.data_ptr())."""

    def __init__(self, n_games: int, num_simulations: int = 250, mcts_batch_size: int = 96, cpuct: float = 1.0,
                 widen_coeff: float = 1.5, dirichlet_alpha: float = 0.1, dirichlet_epsilon: float = 0.25,
                 max_plies: int = 1024, device: int = 0, fast: bool = False, leaves_per_step: int = 8, fast_arena_granules: int = 0):
        self.lib = load_hip_library()
        self.G = int(n_games)
        self.fast, self.L = bool(fast), int(leaves_per_step) if fast else 1
        self.cfg = BoConfig(n_games, num_simulations, mcts_batch_size, max_plies, cpuct, widen_coeff, dirichlet_alpha,
                            dirichlet_epsilon, 1 if fast else 0, self.L, int(fast_arena_granules))
        self.rows = self.G * self.L  # rows of the NN tensors
        self.num_simulations, self.dirichlet_alpha = num_simulations, dirichlet_alpha
        h = C.c_void_p()
        self._check(self.lib.bo_engine_create(C.byref(self.cfg), device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.bo_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(f"betaone engine error {rc}: {self.lib.bo_last_error().decode()}")

    # -- set-up ----------------------------------------------------------------------------------
    @staticmethod
    def _strs(items: Optional[Sequence[Optional[str]]], n: int):
        arr = (C.c_char_p * n)()
        for i in range(n):
            v = items[i] if items is not None else None
            arr[i] = v.encode() if v is not None else None
        return arr

    def reset(self, slots: Sequence[int], fens: Optional[Sequence[Optional[str]]] = None,
              moves: Optional[Sequence[Optional[str]]] = None, stream: int = 0):
        s = _i32(slots)
        n = len(s)
        self._check(self.lib.bo_games_reset(self.h, n, _p(s), self._strs(fens, n), self._strs(moves, n), stream))

    def reset_ex(self, slots, fens, moves, hist: Sequence[Sequence[BoPosition]],
                 trk: Sequence[Sequence[Tuple[BoPosition, int]]], stream: int = 0):
        s = _i32(slots)
        n = len(s)
        harr = (BoPosition * (7 * n))()
        nh = np.zeros(n, dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.int32)
        flat, cnts = [], []
        for i in range(n):
            nh[i] = len(hist[i])
            for k, p in enumerate(hist[i]):
                harr[i * 7 + k] = p
            for p, c in trk[i]:
                flat.append(p)
                cnts.append(c)
            off[i + 1] = len(flat)
        tarr = (BoPosition * max(1, len(flat)))(*flat)
        tc = _i32(cnts if cnts else [0])
        self._check(self.lib.bo_games_reset_ex(self.h, n, _p(s), self._strs(fens, n), self._strs(moves, n), harr, _p(nh),
                                               tarr, _p(tc), _p(off), stream))

    def root_info(self, stream: int = 0):
        nl, tm, ply = (np.zeros(self.G, dtype=np.int32) for _ in range(3))
        self._check(self.lib.bo_root_info(self.h, _p(nl), _p(tm), _p(ply), stream))
        return nl, tm, ply

    # -- search ----------------------------------------------------------------------------------
    def search_begin(self, go, noise: Optional[np.ndarray], nn_in_ptr: int, stream: int = 0):
        g = _i32(go)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=np.float64).reshape(self.G, MAX_LEGAL)
        self._check(self.lib.bo_search_begin(self.h, _p(g), _p(nz, _F64P) if nz is not None else None, nn_in_ptr, stream))

    def step(self, policy_ptr: int, value_ptr: int, kind: int, nn_in_ptr: int, stream: int = 0):
        self._check(self.lib.bo_step(self.h, policy_ptr, value_ptr, kind, nn_in_ptr, stream))

    def step_heads(self, logits_ptr: int, vpart_ptr: int, b1_ptr: int, w2_ptr: int, b2_ptr: int, rows: int, nn_in_ptr: int, stream: int = 0):
        """bo_step with the evaluate stage's tail (row softmax, value head's last layer) inside the step kernel (bo_step_heads)."""
        self._check(self.lib.bo_step_heads(self.h, logits_ptr, vpart_ptr, b1_ptr, w2_ptr, b2_ptr, rows, nn_in_ptr, stream))

    def poll(self, stream: int = 0, want_mask: bool = True):
        run, req = C.c_int32(), C.c_int32()
        mask = np.zeros(self.G, dtype=np.int32) if want_mask else None
        self._check(self.lib.bo_search_poll(self.h, C.byref(run), C.byref(req), _p(mask) if want_mask else None, stream))
        return run.value, req.value, mask

    def search_stop(self, mask=None, stream: int = 0) -> np.ndarray:
        """Interrupt the running searches (all, or those with mask[g] != 0) between two steps; -> simulations completed per game."""
        m = _i32(mask) if mask is not None else None
        done = np.zeros(self.G, dtype=np.int32)
        self._check(self.lib.bo_search_stop(self.h, _p(m) if m is not None else None, _p(done), stream))
        return done

    def result(self, stream: int = 0) -> Dict[str, np.ndarray]:
        G = self.G
        out = dict(n=np.zeros(G, np.int32), idx=np.zeros((G, RES_CAP), np.int32), val=np.zeros((G, RES_CAP), np.float32),
                   best_idx=np.zeros(G, np.int32), best_move=np.zeros(G, np.int32), total=np.zeros(G, np.int32))
        self._check(self.lib.bo_search_result(self.h, _p(out["n"]), _p(out["idx"]), _p(out["val"], _F32P),
                                              _p(out["best_idx"]), _p(out["best_move"]), _p(out["total"]), stream))
        return out

    def play(self, actions, stream: int = 0):
        a = _i32(actions)
        assert a.shape == (self.G,)
        self._check(self.lib.bo_play(self.h, _p(a), stream))

    # -- native per-move host work (RandomState-compatible streams inside the engine) ----------------------
    def rng_seed(self, slot: int, seed: int):
        self._check(self.lib.bo_rng_seed(self.h, slot, seed & 0xFFFFFFFF))

    def rng_get_state(self, slot: int):
        """-> tuple accepted by numpy.random.RandomState.set_state"""
        key = np.zeros(624, dtype=np.uint32)
        pos, hg, g = C.c_int32(), C.c_int32(), C.c_double()
        self._check(self.lib.bo_rng_state(self.h, slot, 0, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                          C.byref(hg), C.byref(g)))
        return ("MT19937", key, pos.value, hg.value, g.value)

    def rng_set_state(self, slot: int, state):
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        pos, hg, g = C.c_int32(int(state[2])), C.c_int32(int(state[3])), C.c_double(float(state[4]))
        self._check(self.lib.bo_rng_state(self.h, slot, 1, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                          C.byref(hg), C.byref(g)))

    def selfplay_sample(self, active, move_number, temperature, out, stream: int = 0):
        """out: dict of preallocated arrays n[G], idx[G,RES_CAP], val[G,RES_CAP], best_idx[G], action[G]."""
        a, m = _i32(active), _i32(move_number)
        th, ti, tf = temperature
        self._check(self.lib.bo_selfplay_sample(self.h, _p(a), _p(m), int(th), float(ti), float(tf), _p(out["n"]),
                                                _p(out["idx"]), _p(out["val"], _F32P), _p(out["best_idx"]),
                                                _p(out["action"]), stream))
        return out

    def selfplay_noise(self, stream: int = 0):
        self._check(self.lib.bo_selfplay_noise(self.h, stream))

    def selfplay_begun(self):
        """(n_legal, terminal, go) of the searches a selfplay_turn(lazy_begin=True) began; to be called after the first
        evaluation's forward has been enqueued and before selfplay_noise."""
        nl, tm, go = (np.zeros(self.G, dtype=np.int32) for _ in range(3))
        self._check(self.lib.bo_selfplay_begun(self.h, _p(nl), _p(tm), _p(go)))
        return nl, tm, go

    def result_prefetch(self, stream: int = 0) -> None:
        """Enqueue the result block's trip to pinned host memory behind the searches (no wait): selfplay_turn(prefetched=True) then
        needs no round trip of its own.  Stale after any further step of the searches."""
        self._check(self.lib.bo_search_result_prefetch(self.h, stream))

    def selfplay_turn(self, active, move_number, temperature, out, want_next, nn_in_ptr: int, stream: int = 0, defer_noise: bool = False,
                      poll_first: bool = False, lazy_begin: bool = False, prefetched: bool = False):
        """selfplay_sample + play + selfplay_begin(want_next) in one call.  Returns (out, (n_legal, terminal, go) or None):
        None when a game needs the dense NumPy sampler (action -3) -- nothing was played then.  poll_first: check that all
        searches are finished first; returns (None, None) if one is still running (issue another step and call again).
        lazy_begin (with defer_noise): the begin does not wait for the device; the second item is then LAZY_BEGIN and
        selfplay_begun() returns the triple later."""
        a, m, w = _i32(active), _i32(move_number), _i32(want_next)
        th, ti, tf = temperature
        nl, tm, go = (np.zeros(self.G, dtype=np.int32) for _ in range(3))
        done = C.c_int32(0)
        self._check(self.lib.bo_selfplay_turn(self.h, _p(a), _p(m), int(th), float(ti), float(tf), _p(out["n"]), _p(out["idx"]),
                                              _p(out["val"], _F32P), _p(out["best_idx"]), _p(out["action"]), _p(w), nn_in_ptr, _p(nl), _p(tm),
                                              _p(go), (1 if defer_noise else 0) | (2 if poll_first else 0) | (4 if lazy_begin else 0) | (8 if (prefetched and poll_first) else 0),
                                              C.byref(done), stream))
        if done.value < 0:
            return None, None
        if done.value == 2:
            return out, LAZY_BEGIN
        return out, ((nl, tm, go) if done.value else None)

    def selfplay_autoturn(self, active, move_number, temperature, want_next, nn_in_ptr: int, stream: int = 0, redo: bool = False) -> None:
        """The ply's turn enqueued on the device behind the searches (bo_selfplay_autoturn): sample, play, begin the next searches;
        the moves' uniforms are drawn now.  Nothing waits; autoturn_collect returns what happened."""
        a, m, w = _i32(active), _i32(move_number), _i32(want_next)
        th, ti, tf = temperature
        self._check(self.lib.bo_selfplay_autoturn(self.h, _p(a), _p(m), int(th), float(ti), float(tf), _p(w), nn_in_ptr, 1 if redo else 0, stream))

    def autoturn_supported(self, temperature) -> bool:
        """The settings bo_selfplay_autoturn covers: the reference's search semantics with a root of <= 2 children, T_initial = 1, T_final > 0."""
        _, ti, tf = temperature
        if os.environ.get("BETAONE_TURN_COPIES", "0") == "1":  # (the A/B switch that moves the turn's blocks by copy commands: host turn only)
            return False
        return (not self.fast) and int(self.cfg.widen_coeff) == 1 and self.cfg.num_simulations >= 1 and abs(float(ti) - 1.0) < 1e-6 and float(tf) > 0.0

    def autoturn_ready(self) -> bool:
        r = C.c_int32(0)
        self._check(self.lib.bo_selfplay_autoturn_ready(self.h, C.byref(r)))
        return bool(r.value)

    def autoturn_collect(self, out):
        """(out, (n_legal, terminal, go)) of the device's turn, or (None, None) if a search was still running when it came up (nothing
        was played: step once more and enqueue selfplay_autoturn(redo=True))."""
        nl, tm, go = (np.zeros(self.G, dtype=np.int32) for _ in range(3))
        done = C.c_int32(0)
        self._check(self.lib.bo_selfplay_autoturn_collect(self.h, _p(out["n"]), _p(out["idx"]), _p(out["val"], _F32P), _p(out["best_idx"]),
                                                          _p(out["action"]), _p(nl), _p(tm), _p(go), C.byref(done)))
        if done.value < 0:
            return None, None
        return out, (nl, tm, go)

    def selfplay_begin(self, want, nn_in_ptr: int, stream: int = 0):
        w = _i32(want)
        nl, tm, go = (np.zeros(self.G, dtype=np.int32) for _ in range(3))
        self._check(self.lib.bo_selfplay_begin(self.h, _p(w), nn_in_ptr, _p(nl), _p(tm), _p(go), stream))
        return nl, tm, go

    # -- records / introspection --------------------------------------------------------------------
    def export_game(self, slot: int, stream: int = 0, n_plies: Optional[int] = None):
        """(positions[n+1], moves[n]) of the game in `slot`; n_plies = the caller's own count sizes the buffers."""
        cap = (self.cfg.max_plies if n_plies is None else min(int(n_plies), self.cfg.max_plies)) + 1
        pos = (BoPosition * cap)()
        mv = np.zeros(cap, dtype=np.int32)
        n = C.c_int32()
        self._check(self.lib.bo_game_export(self.h, slot, pos, _p(mv), cap, C.byref(n), stream))
        return PositionList(pos, n.value + 1), mv[:n.value].tolist()

    def encode_game(self, slot: int, first: int, n: int, out_ptr: int, stream: int = 0):
        self._check(self.lib.bo_game_encode(self.h, slot, first, n, out_ptr, stream))

    def debug_tree(self, slot: int, stream: int = 0) -> List[dict]:
        n = C.c_int32()
        self._check(self.lib.bo_debug_tree(self.h, slot, None, 0, C.byref(n), stream))
        arr = (BoNode * max(1, n.value))()
        self._check(self.lib.bo_debug_tree(self.h, slot, arr, n.value, C.byref(n), stream))
        return [dict(parent=a.parent, n=a.n_visits, q=np.float32(a.q_value), prior=np.float32(a.prior), move=a.move,
                     n_children=a.n_children, first_child=a.first_child, terminal=a.terminal) for a in arr[:n.value]]

    def status(self, stream: int = 0) -> Dict[str, np.ndarray]:
        names = ["status", "evals", "flushes", "term_sims", "levels", "children_scanned"]
        out = {k: np.zeros(self.G, np.int32) for k in names}
        self._check(self.lib.bo_engine_status(self.h, *[_p(out[k]) for k in names], stream))
        return out

    def fast_options(self, tree_reuse=None, games_per_halfwave=None, select_flags=None):
        """FAST mode knobs (None = unchanged): tree reuse between moves; games the select + backup kernel interleaves per
        half-wavefront (2 or 4); select_flags = SEL_NT | SEL_ROOT_IN_REGS | SEL_DENSE (include/betaone_engine.h)."""
        self._check(self.lib.bo_fast_options(self.h, -1 if tree_reuse is None else (1 if tree_reuse else 0),
                                             -1 if games_per_halfwave is None else int(games_per_halfwave),
                                             -1 if select_flags is None else int(select_flags)))

    SEL_NT, SEL_ROOT_IN_REGS, SEL_DENSE, SEL_LANE, SEL_OCT, SEL_QUAD = 1, 2, 4, 8, 16, 32
    def debug_fast(self, slot: int, stream: int = 0):
        """(control block of game `slot` as a dict of its fields, paths [L, 64] of the step's simulations) -- bo_debug_fast."""
        L = self.L
        cs = (16 + 7 * L + 31) // 32 * 32
        ctl, paths = np.zeros(cs, np.int32), np.zeros((L, 64), np.int32)
        self._check(self.lib.bo_debug_fast(self.h, slot, _p(ctl), cs, _p(paths), stream))
        names = ["row_slot", "row_plink", "row_nlegal", "row_term", "row_sim", "sim_row", "sim_plen"]
        out = dict(n_rows=int(ctl[0]), n_step=int(ctl[1]), cur=int(ctl[2]), top=int(ctl[3]), levels=int(ctl[4]), kids=int(ctl[5]),
                   granules=int(ctl[6]), path_nodes=int(ctl[7]), term_sims=int(ctl[8]))
        for i, nm in enumerate(names):
            out[nm] = ctl[16 + i * L:16 + (i + 1) * L].copy()
        return out, paths

    GRANULE_BYTES = 128  # BO_FAST_GRANULE_BYTES

    def fast_stats(self, stream: int = 0, time_select: int = -1) -> Dict[str, np.ndarray]:
        """granules_read / arena_granules per game (x GRANULE_BYTES = bytes); select_ms / select_launches of the timed select +
        backup kernel since time_select=1 (eager steps only); time_select: 1 on, 0 off, -1 unchanged."""
        gran, pnodes, top = np.zeros(self.G, np.uint64), np.zeros(self.G, np.uint64), np.zeros(self.G, np.int32)
        ms, n = C.c_double(0.0), C.c_int64(0)
        u64 = C.POINTER(C.c_uint64)
        self._check(self.lib.bo_fast_stats(self.h, gran.ctypes.data_as(u64), pnodes.ctypes.data_as(u64), _p(top), time_select,
                                           C.byref(ms), C.byref(n), stream))
        return dict(granules_read=gran, path_nodes=pnodes, arena_granules=top, select_ms=ms.value, select_launches=n.value)

    def event_pair_overhead_ms(self, samples: int = 32, stream: int = 0) -> float:
        """What a HIP event pair around one kernel launch measures beyond the kernel (median over `samples` empty launches), in ms."""
        ms = C.c_double(0.0)
        self._check(self.lib.bo_event_pair_overhead(C.byref(ms), samples, stream))
        return ms.value

    def watch(self, dev_word_ptr: int, n_words: int = 1):
        """Have every fetched result block bring the int32 device word(s) at `dev_word_ptr` along (0: none) -- bo_engine_watch_words."""
        self._check(self.lib.bo_engine_watch_words(self.h, dev_word_ptr or None, int(n_words)))

    def watch_seen(self, clear: bool = True) -> int:
        seen = C.c_int32(0)
        self._check(self.lib.bo_engine_watch_seen(self.h, C.byref(seen), 1 if clear else 0))
        return seen.value

    def status_bits(self, stream: int = 0) -> np.ndarray:
        """Only the per-slot status words (one small copy): 0 = fine, else a combination of STATUS_BITS."""
        st = np.zeros(self.G, np.int32)
        self._check(self.lib.bo_engine_status(self.h, _p(st), None, None, None, None, None, stream))
        return st

    @staticmethod
    def describe_status(bits: int) -> str:
        return ", ".join(v for b, v in STATUS_BITS.items() if bits & b)

    def soft_status_bits(self) -> int:
        """Status bits that are conditions of ONE search rather than faults of the engine (fast mode: a full arena)."""
        return ST_NODE_OVERFLOW if self.fast else 0

    def check_status(self) -> int:
        """Raises on an engine fault in any slot; returns the number of slots that carry only soft conditions (fast mode: slots
        whose arena was full at some point -- see ST_NODE_OVERFLOW)."""
        st = self.status_bits()
        bad = np.nonzero(st & ~self.soft_status_bits())[0]
        if len(bad):
            g = int(bad[0])
            raise EngineError(f"game slot {g}: {self.describe_status(int(st[g]))}")
        return int(np.count_nonzero(st))

    def profile(self, enable: int = -1, read: bool = True, stream: int = 0):
        out = np.zeros((self.G, PROF_SLOTS), dtype=np.uint64) if read else None
        self._check(self.lib.bo_debug_profile(self.h, enable, out.ctypes.data_as(C.POINTER(C.c_uint64)) if read else None, stream))
        return out

    def movegen(self, positions: Sequence[BoPosition], stream: int = 0):
        n = len(positions)
        arr = (BoPosition * n)(*positions)
        mv = np.zeros((n, MAX_LEGAL), np.int32)
        cnt, chk = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._check(self.lib.bo_movegen_batch(self.h, n, arr, _p(mv), _p(cnt), _p(chk), stream))
        return [[int(m) for m in mv[i, :cnt[i]]] for i in range(n)], chk
