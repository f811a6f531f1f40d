"""
self_play -- `run_self_play_game(model, game_id)` / `save_game_data` of
/root/reference/self_play.py:84-231 over the MI355X engine, plus the batched entry point the engine is
built for (`run_self_play_games`: thousands of games in one process, one RandomState per game).

Return type is the reference's: a list of `(torch.FloatTensor[120,8,8], np.ndarray float32[4672], float)`
per ply, or None when the game was aborted (self_play.py:119,167,180); main.py and train.py consume it
unchanged (pickle format of save_game_data, train.py:187-219).
"""
import os
import pickle
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

import config
from betaone_amd import engine as E
from betaone_amd import sampling
from betaone_amd.rollout import CohortRollout, FinishedGame, Rollout

SelfPlayData = Tuple[torch.Tensor, np.ndarray, float]

def apply_temperature(probs: np.ndarray, temperature: float) -> np.ndarray:
    return sampling.apply_temperature(probs, temperature, np.random)


def select_move_with_temperature(probs: np.ndarray, move_number: int) -> int:
    return sampling.select_move_with_temperature(probs, move_number, np.random, config.TEMPERATURE_THRESHOLD,
                                                 config.TEMPERATURE_INITIAL, config.TEMPERATURE_FINAL)


def _inference_copy(model, n_slots: int):
    dev = E.runtime_device(config.DEVICE)  # raises off the GPU: there is no CPU path
    if dev.type == "cuda" and hasattr(model, "for_inference"):  # BN-folded copy on the hand-written evaluate stage for n_slots rows
        from betaone_amd.nn_tune import best_inference_copy

        rows = n_slots * (config.FAST_LEAVES if config.SEARCH_MODE == "fast" else 1)
        model = best_inference_copy(model, rows, dev, next(model.parameters()).dtype)
    return model


def _cohorts(n_slots: int, rng_mode: str) -> int:
    k = int(getattr(config, "COHORTS", 1))
    floor = int(getattr(config, "COHORT_MIN_SLOTS", 64))  # (tests lower it to drive small cohorts)
    if rng_mode != "native" or config.SEARCH_MODE == "fast":
        return 1
    while k > 1 and (n_slots % k or n_slots // k < floor):  # COHORTS, COHORTS/2, ...: the largest that fits
        k //= 2
    return max(k, 1)


def _rollout(model, n_slots: int, rng_mode: str = "python"):
    dev = E.runtime_device(config.DEVICE)
    k = _cohorts(n_slots, rng_mode)
    model = _inference_copy(model, n_slots // k)
    kw = dict(num_simulations=config.NUM_SIMULATIONS, mcts_batch_size=config.MCTS_BATCH_SIZE,
              cpuct=config.CPUCT, widen_coeff=config.WIDEN_COEFF, dirichlet_alpha=config.DIRICHLET_ALPHA,
              dirichlet_epsilon=config.DIRICHLET_EPSILON, max_plies=config.ENGINE_MAX_PLIES,  # None = room for MAX_GAME_MOVES
              max_game_moves=config.MAX_GAME_MOVES,
              temperature=(config.TEMPERATURE_THRESHOLD, config.TEMPERATURE_INITIAL, config.TEMPERATURE_FINAL),
              device=dev, autocast=config.AUTOCAST, rng_mode=rng_mode,
              policy_kind="probs" if config.POLICY_SOFTMAX == "torch" else "logits",
              fast=config.SEARCH_MODE == "fast", leaves_per_step=config.FAST_LEAVES)
    return CohortRollout(model, n_slots, cohorts=k, **kw) if k > 1 else Rollout(model, n_slots, **kw)


def _records(ro: Rollout, fin: FinishedGame) -> List[SelfPlayData]:
    n = len(fin.pis)
    states = ro.encode_finished_in_slot(fin.slot, n, fin.first_ply).cpu()
    out = []
    for i in range(n):
        idx, val = fin.pis[i]
        out.append((states[i].clone(), sampling.dense_pi(idx, val), fin.z(i)))
    return out


def run_self_play_games(model, game_ids: Sequence[int], seeds: Optional[Sequence[int]] = None,
                        n_slots: Optional[int] = None, start_fens: Optional[Sequence[Optional[str]]] = None,
                        on_game=None, dense: bool = True, reload_model=None, on_records=None
                        ) -> Dict[int, Optional[List[SelfPlayData]]]:
    """Play len(game_ids) games, n_slots at a time, on one GPU.  Game i draws its Dirichlet noise and its
    moves from numpy.random.RandomState(seeds[i]) -- the stream the reference consumes after
    np.random.seed(seeds[i]) -- so results do not depend on n_slots or on which GPU a game lands on.
    on_game(FinishedGame): called for every finished game while it is still resident in its slot (compact records:
    betaone_amd.records.save_games); dense=False skips the reference's dense tuples (the result values are then empty lists).
    on_records(game_id, tuples) -> what to keep in the result for that game: called with a finished game's dense tuples as soon
    as they exist (selfplay_main pickles them there and keeps an empty list, so an iteration's tuples never pile up in memory).
    reload_model() -> None | a new PolicyValueNet: polled once per ply; a returned model replaces the evaluate stage for every
    evaluation from the next ply on (main.py:147-148 hands weights to its workers through best_model.pth: betaone_amd.selfplay_main
    watches that file)."""
    ids = list(game_ids)
    seeds = list(seeds) if seeds is not None else ids
    n_slots = min(n_slots or len(ids), len(ids))
    ro = _rollout(model, n_slots, rng_mode="native")  # RandomState(seed)-compatible streams kept inside the engine
    results: Dict[int, Optional[List[SelfPlayData]]] = {}
    queue = list(range(len(ids)))

    def next_game(_slot):
        if not queue:
            return None
        i = queue.pop(0)
        return ids[i], int(seeds[i]), (start_fens[i] if start_fens else None)

    def finished(fin: FinishedGame):
        if fin.terminal == 0:
            print(f"Game {fin.game_id} aborted after {len(fin.moves)} moves (max).")  # self_play.py:186-187
        if on_game is not None:
            on_game(fin)
        data = _records(ro, fin) if dense else []
        results[fin.game_id] = on_records(fin.game_id, data) if (on_records is not None and dense) else data

    first = [next_game(s) for s in range(n_slots)]
    ro.start_games(list(range(n_slots)), [f[0] for f in first], [f[1] for f in first], [f[2] for f in first])
    try:
        while any(g is not None for g in ro.games):  # (cohorts: a call ends each cohort's outstanding ply and begins its next one)
            if reload_model is not None:
                fresh = reload_model()
                if fresh is not None:
                    ro.swap_model(_inference_copy(fresh, n_slots // _cohorts(n_slots, "native")))
            ro.play_ply(on_finished=finished, refill=next_game)
            _settle_status(ro, results, finished, next_game)
    finally:
        ro.close()
    return results


def _settle_status(ro: Rollout, results, finished, next_game):
    """Per-game conditions end THAT game only, as in the reference (one game per call there): an unplayable move is the
    abort of self_play.py:167 (-> None); a slot whose position stack is full (config.ENGINE_MAX_PLIES smaller than
    MAX_GAME_MOVES) is the move-limit stop of self_play.py:186 (records kept).  Anything else is an engine fault."""
    st = ro.eng.status_bits()
    soft = E.ST_NODE_OVERFLOW if ro.fast else 0
    for g in np.nonzero(st)[0]:
        g, bits = int(g), int(st[g])
        if ro.games[g] is None:
            continue
        bits &= ~soft  # (fast mode: a full arena narrows that search, it does not end the game)
        if not bits:
            continue
        if bits & ~(E.ST_PLY_OVERFLOW | E.ST_ILLEGAL_ACTION):
            raise E.EngineError(f"game {ro.games[g].game_id} (slot {g}): {ro.eng.describe_status(bits)}")
        if bits & E.ST_ILLEGAL_ACTION:
            results[ro.games[g].game_id] = None
            ro.retire(g, None, next_game)
        else:
            ro.retire(g, finished, next_game)


def run_self_play_game(model, game_id: int) -> Optional[List[SelfPlayData]]:
    """self_play.py:84-216: one game from the standard start position, RNG = numpy's process-global
    legacy generator exactly as in the reference (mcts.py:192, self_play.py:73)."""
    ro = _rollout(model, 1)
    out: List[Optional[List[SelfPlayData]]] = [None]

    def finished(fin: FinishedGame):
        if fin.terminal == 0:
            print(f"Game {game_id} aborted after {len(fin.moves)} moves (max).")
        out[0] = _records(ro, fin)

    ro.start_games([0], [game_id], [np.random])
    results = {}
    try:
        while ro.games[0] is not None:
            ro.play_ply(on_finished=finished)
            _settle_status(ro, results, finished, None)
    finally:
        ro.close()
    return out[0]


play_game = run_self_play_game  # north-star alias (BASELINE.json)


def save_game_data(game_data: List[SelfPlayData], iteration: int, game_id: int):
    """self_play.py:220-231: pickle to DATA_DIR/iter_{iteration}/game_{game_id}.pkl."""
    if not game_data:
        return
    data_dir = os.path.join(config.DATA_DIR, f"iter_{iteration}")
    os.makedirs(data_dir, exist_ok=True)
    filepath = os.path.join(data_dir, f"game_{game_id}.pkl")
    try:  # (through a temporary name: a killed writer leaves no half pickle that a resume would count as a finished game)
        with open(filepath + ".tmp", "wb") as f:
            pickle.dump(game_data, f)
        os.replace(filepath + ".tmp", filepath)
    except Exception as e:
        print(f"Error saving game data to {filepath}: {e}")
