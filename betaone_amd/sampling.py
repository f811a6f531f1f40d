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
    """self_play.py:25-56, literal."""
    if temperature == 0:
        new_probs = np.zeros_like(probs)
        max_prob_indices = np.where(probs == np.max(probs))[0]
        if len(max_prob_indices) == 0:
            return new_probs
        chosen_index = rng.choice(max_prob_indices)
        new_probs[chosen_index] = 1.0
        return new_probs
    elif abs(temperature - 1.0) < 1e-6:
        return probs
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            scaled_probs = np.power(probs.astype(np.float64), 1.0 / temperature)
        scaled_probs[~np.isfinite(scaled_probs)] = 0.0
        sum_scaled_probs = np.sum(scaled_probs)
        if sum_scaled_probs > 1e-9:
            normalized_probs = (scaled_probs / sum_scaled_probs).astype(np.float32)
            renorm_sum = np.sum(normalized_probs)
            if abs(renorm_sum - 1.0) > 1e-6 and renorm_sum > 1e-9:
                normalized_probs /= renorm_sum
            return normalized_probs
        else:
            non_zero_indices = np.where(probs > 1e-9)[0]
            num_non_zero = len(non_zero_indices)
            if num_non_zero > 0:
                uniform_probs = np.zeros_like(probs, dtype=np.float32)
                uniform_probs[non_zero_indices] = 1.0 / num_non_zero
                return uniform_probs
            else:
                return probs.astype(np.float32)


def select_move_with_temperature(probs: np.ndarray, move_number: int, rng=np.random, threshold: int = 30,
                                 t_initial: float = 1.0, t_final: float = 0.1) -> int:
    """self_play.py:59-80, literal (dense)."""
    temp = t_initial if move_number < threshold else t_final
    temp_scaled_probs = apply_temperature(probs, temp, rng)
    try:
        prob_sum = np.sum(temp_scaled_probs)
        if abs(prob_sum - 1.0) > 1e-6:
            if prob_sum > 1e-9:
                temp_scaled_probs /= prob_sum
            else:
                return int(np.argmax(probs))
        action_index = rng.choice(len(temp_scaled_probs), p=temp_scaled_probs)
    except ValueError as e:
        print(f"Error sampling move: {e}\nFalling back to argmax of original probabilities.")
        action_index = np.argmax(probs)
    return int(action_index)


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
