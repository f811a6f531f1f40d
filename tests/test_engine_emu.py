"""CPU tests of the ENGINE'S DEVICE CODE (betaone_amd/csrc) under the 64-lane wave emulator: the
same sources hipcc compiles for gfx950, executed lane by lane on the CPU.  They check kernel logic
(and memory safety) before any GPU minute is spent; the `-m gpu` tests repeat them on the product
library.  Nothing here is a product path."""
import numpy as np
import pytest

import engine_cases as EC


@pytest.fixture(scope="module")
def backend():
    return "emu"


def test_movegen_matches_oracle_order(backend):
    EC.check_movegen_random_positions(backend, n_games=12, max_plies=60, seed=0)


def test_movegen_special_positions(backend):
    EC.check_movegen_special(backend)


@pytest.mark.parametrize("name", EC.SEARCH_NAMES)
def test_search_matches_reference_trace(backend, name):
    EC.check_golden_search(backend, name)


@pytest.mark.parametrize("name", EC.GAME_NAMES)
def test_self_play_game_matches_reference(backend, name):
    EC.check_golden_game(backend, name)


def test_many_games_in_lockstep_match_oracle(backend):
    EC.check_multi_game_vs_oracle(backend, n_games=5, plies=6, sims=60, batch=16)


def test_full_games_to_termination_match_oracle(backend):
    """Whole games until is_game_over(claim_draw=True): long-game logic (claimable draws in the REAL game, tracker and
    position-stack growth, end-of-game tracker in the training encodings) against the oracle."""
    EC.check_full_games_vs_oracle(backend, n_games=4, sims=12, batch=8)


def test_edge_cases_maximum_sizes_and_error_paths(backend):
    EC.check_edge_cases(backend)


@pytest.mark.parametrize("cfg", EC.SEARCH_CONFIG_SWEEP, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_search_configuration_sweep_matches_oracle(backend, cfg):
    """Batch sizes around the simulation count, unusual CPUCT / widening / Dirichlet settings: moves, pi and states bit-exact."""
    EC.check_multi_game_vs_oracle(backend, n_games=2, plies=3, **cfg)


def test_games_from_special_start_positions_match_oracle(backend):
    EC.check_games_from_positions_vs_oracle(backend)


@pytest.mark.parametrize("case", EC.LONG_TERMINAL_RUN_CASES, ids=lambda c: f"{c[0].split()[0][:12]}-{c[2]}")
def test_long_runs_of_terminal_simulations_match_oracle(backend, case):
    EC.check_long_terminal_runs_vs_oracle(backend, case)


def test_watched_status_word_arrives_with_the_result_block(backend):
    EC.check_watched_status_word(backend)
