"""
betaone_amd/sampling.py -- the per-move host work of self-play, bit-compatible with the reference's NumPy calls.

  np.random.dirichlet([alpha]*n)                          /root/reference/mcts.py:192
  apply_temperature / select_move_with_temperature        /root/reference/self_play.py:25-80
  np.random.choice(4672, p=...)                           /root/reference/self_play.py:73

`rng` is either the module `numpy.random` (the reference's process-global legacy RNG; what the
single-game drop-in uses) or a `numpy.random.RandomState(seed)` (one independent legacy stream per
game when thousands of games run in one process).  Both expose the same legacy generator, so a game
driven by RandomState(s) consumes exactly the stream the reference consumes after np.random.seed(s).

The search result is sparse (the reference's root has <= 2 children), so the hot
path never materialises 4672-vectors: `select_action_sparse` reproduces the dense computation
exactly for <= 2 non-zero entries (every float operation on the zeros is exact) and falls back to the
literal dense mirror otherwise.
"""
from __future__ import annotations

import numpy as np

NUM_ACTIONS = 4672


def apply_temperature(probs: np.ndarray, temperature: float, rng=np.random) -> np.ndarray:
    """self_play.py:25-56: the same NumPy operations in the same order (argmax one-hot for T == 0, identity for
    T == 1, otherwise p**(1/T) in float64, non-finite -> 0, normalise, cast to float32, re-normalise if the
    float32 sum is off by more than 1e-6; uniform over the support when everything underflowed)."""
    if temperature == 0:
        onehot = np.zeros_like(probs)
        ties = np.where(probs == np.max(probs))[0]
        if len(ties):
            onehot[rng.choice(ties)] = 1.0
        return onehot
    if abs(temperature - 1.0) < 1e-6:
        return probs
    with np.errstate(divide="ignore", invalid="ignore"):
        powered = np.power(probs.astype(np.float64), 1.0 / temperature)
    powered[~np.isfinite(powered)] = 0.0
    total = np.sum(powered)
    if total > 1e-9:
        out = (powered / total).astype(np.float32)
        check = np.sum(out)
        if abs(check - 1.0) > 1e-6 and check > 1e-9:
            out /= check
        return out
    support = np.where(probs > 1e-9)[0]
    if len(support) == 0:
        return probs.astype(np.float32)
    flat = np.zeros_like(probs, dtype=np.float32)
    flat[support] = 1.0 / len(support)
    return flat


def select_move_with_temperature(probs: np.ndarray, move_number: int, rng=np.random, threshold: int = 30,
                                 t_initial: float = 1.0, t_final: float = 0.1) -> int:
    """self_play.py:59-80 (dense): temperature by full-move number, one legacy `choice` draw, argmax fallbacks."""
    p = apply_temperature(probs, t_initial if move_number < threshold else t_final, rng)
    try:
        mass = np.sum(p)
        if abs(mass - 1.0) > 1e-6:
            if not mass > 1e-9:
                return int(np.argmax(probs))
            p /= mass
        return int(rng.choice(len(p), p=p))
    except ValueError as err:
        print(f"Error sampling move: {err}; falling back to the most visited move")
        return int(np.argmax(probs))


def dense_pi(idx: np.ndarray, val: np.ndarray) -> np.ndarray:
    pi = np.zeros(NUM_ACTIONS, dtype=np.float32)
    pi[idx] = val
    return pi


def select_action_sparse(idx: np.ndarray, val: np.ndarray, move_number: int, rng=np.random, threshold: int = 30,
                         t_initial: float = 1.0, t_final: float = 0.1) -> int:
    """select_move_with_temperature on a sparse pi given as (action indices, float32 values)."""
    n = len(idx)
    temp = t_initial if move_number < threshold else t_final
    if n == 1 and temp != 0 and val[0] == 1.0:
        # one-hot pi: every temperature leaves [1.0]; choice() still draws its one uniform (cdf = [1.0] > u)
        rng.random_sample()
        return int(idx[0])
    if n == 0 or n > 2 or temp == 0:
        return select_move_with_temperature(dense_pi(idx, val), move_number, rng, threshold, t_initial, t_final)
    order = np.argsort(idx, kind="stable")
    idx = np.asarray(idx)[order]
    p = np.asarray(val, dtype=np.float32)[order]
    if not abs(temp - 1.0) < 1e-6:  # self_play.py:37-45 on the non-zero entries
        with np.errstate(divide="ignore", invalid="ignore"):
            scaled = np.power(p.astype(np.float64), 1.0 / temp)
        scaled[~np.isfinite(scaled)] = 0.0
        s = np.sum(scaled)
        if not s > 1e-9:
            return select_move_with_temperature(dense_pi(idx, p), move_number, rng, threshold, t_initial, t_final)
        p = (scaled / s).astype(np.float32)
        rs = np.sum(p)
        if abs(rs - 1.0) > 1e-6 and rs > 1e-9:
            p = p / rs
    prob_sum = np.sum(p)  # self_play.py:68-73
    if abs(prob_sum - 1.0) > 1e-6:
        if prob_sum > 1e-9:
            p = p / prob_sum
        else:
            return int(idx[int(np.argmax(val[order]))])
    # RandomState.choice(a, p=p): cdf = p.astype(double).cumsum(); cdf /= cdf[-1]; searchsorted(random_sample(), 'right')
    pd = p.astype(np.float64)
    if abs(float(np.sum(pd)) - 1.0) > 3.5e-4 or (pd < 0).any() or np.isnan(pd).any():
        return select_move_with_temperature(dense_pi(idx, val[order]), move_number, rng, threshold, t_initial, t_final)
    cdf = pd.cumsum()
    cdf /= cdf[-1]
    u = rng.random_sample()
    k = int(cdf.searchsorted(u, side="right"))
    return int(idx[min(k, n - 1)])


def root_noise(n_legal: int, alpha: float, rng=np.random) -> np.ndarray:
    """mcts.py:192."""
    return rng.dirichlet([alpha] * int(n_legal))
