"""GPU parity tests (run with -m gpu on an MI355X): the product library csrc/libbetaone_hip.so, called
through the C ABI, against the golden traces of the reference and against the CPU oracle."""
import numpy as np
import pytest

import engine_cases as EC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from betaone_amd import engine as E

    E.load_hip_library()  # fails loudly if the HIP library is missing
    return "hip"


def test_movegen_matches_oracle_order(backend):
    EC.check_movegen_random_positions(backend, n_games=40, max_plies=120, seed=1)


def test_movegen_special_positions(backend):
    EC.check_movegen_special(backend)


@pytest.mark.parametrize("name", EC.SEARCH_NAMES)
def test_search_matches_reference_trace(backend, name):
    EC.check_golden_search(backend, name)


@pytest.mark.parametrize("name", EC.GAME_NAMES)
def test_self_play_game_matches_reference(backend, name):
    EC.check_golden_game(backend, name)


def test_many_games_in_lockstep_match_oracle(backend):
    EC.check_multi_game_vs_oracle(backend, n_games=24, plies=10, sims=120, batch=32)
