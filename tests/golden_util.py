"""tests/golden_util.py -- loads tests/golden/* and rebuilds the contexts the fixtures describe."""
from __future__ import annotations

import json
import os

import numpy as np

from fake_model import planes_key

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
NUM_ACTIONS = 4672


class SeamTable:
    """(encoded position, fake-model id) -> (softmax probs, value) exactly as the reference consumed
    them when the fixtures were generated (tests/golden/generate_golden.py)."""

    def __init__(self):
        # g2: the round-1 searches and short games; g5: the two full-length games (generate_golden.py --long)
        zs = [np.load(os.path.join(GOLD, f)) for f in ("g2_evals.npz", "g5_long_evals.npz")]
        cat = lambda key: np.concatenate([z[key] for z in zs])
        self.full = {k: i for i, k in enumerate(cat("full_keys").tolist())}
        self.full_probs, self.full_values = cat("full_probs"), cat("full_values")
        self.sparse = {k: i for i, k in enumerate(cat("sparse_keys").tolist())}
        self.idx, self.val = cat("sparse_idx"), cat("sparse_val")
        self.sparse_values = cat("sparse_values")
        ptrs, base = [], 0
        for z in zs:  # (row pointers of the second table continue behind the first table's entries)
            p = z["sparse_ptr"]
            ptrs.append(p[:-1] + base)
            base += int(p[-1])
        self.ptr = np.concatenate(ptrs + [np.array([base], np.int64)])
        self.misses = []

    def lookup(self, planes_row: np.ndarray, scale: float, salt: int):
        k = f"{planes_key(planes_row)}:{scale}:{salt}"
        if k in self.full:
            i = self.full[k]
            return self.full_probs[i], self.full_values[i]
        if k in self.sparse:
            i = self.sparse[k]
            p = np.zeros(NUM_ACTIONS, dtype=np.float32)
            a, b = self.ptr[i], self.ptr[i + 1]
            p[self.idx[a:b]] = self.val[a:b]
            return p, self.sparse_values[i]
        self.misses.append(k)
        raise KeyError(f"position not in the golden seam table (encoding differs from the reference?): {k}")

    def eval_fn(self, scale: float, salt: int):
        def fn(planes: np.ndarray):
            n = planes.shape[0]
            probs = np.zeros((n, NUM_ACTIONS), dtype=np.float32)
            vals = np.zeros(n, dtype=np.float32)
            for i in range(n):
                probs[i], vals[i] = self.lookup(planes[i], scale, salt)
            return probs, vals

        return fn


def load_searches():
    return json.load(open(os.path.join(GOLD, "g2_searches.json")))


def load_games():
    return json.load(open(os.path.join(GOLD, "g2_games.json"))) + json.load(open(os.path.join(GOLD, "g5_long_games.json")))


def load_codec():
    return json.load(open(os.path.join(GOLD, "g4_codec.json")))


def oracle_cfg(O, c):
    return O.default_config(**c)


def f32bits(x) -> int:
    return int(np.float32(x).view(np.uint32))
