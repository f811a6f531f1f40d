"""GPU tests of the drop-in functions on the product path (cuda:0, libbetaone_hip.so, hipGraph step)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    import torch

    assert torch.cuda.is_available()
    sys.path.insert(0, os.path.join(ROOT, "oracle", "shim"))
    from betaone_amd import dropin

    dropin.install()
    import chess, config, mcts, network, self_play, utils

    assert config.DEVICE == "cuda"
    saved = (config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS, config.NUM_SIMULATIONS, config.MAX_GAME_MOVES)
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 3, 1, 64
    torch.manual_seed(0)
    model = network.PolicyValueNet().to("cuda").eval()
    yield dict(chess=chess, config=config, mcts=mcts, self_play=self_play, utils=utils, model=model, torch=torch)
    (config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS, config.NUM_SIMULATIONS, config.MAX_GAME_MOVES) = saved


def test_run_mcts_on_gpu_is_deterministic_and_well_formed(mods):
    chess, config, mcts, utils, model = mods["chess"], mods["config"], mods["mcts"], mods["utils"], mods["model"]
    config.NUM_SIMULATIONS = 300
    board = chess.Board()
    tracker = utils.RepetitionTracker()
    tracker.add_board(board)
    hist = [board.copy()]
    for u in "d2d4 g8f6 c2c4".split():
        board.push(chess.Move.from_uci(u)); tracker.add_board(board); hist.append(board.copy())
    history = hist[max(0, len(hist) - 8):-1]
    outs = []
    for _ in range(3):  # first call builds the tuned net + captures the hipGraph, later calls replay it
        np.random.seed(5)
        best, pi = mcts.run_mcts(board, model, history, tracker)
        outs.append((best.uci(), pi.copy()))
    assert all(o[0] == outs[0][0] and np.array_equal(o[1], outs[0][1]) for o in outs)
    best, pi = outs[0]
    assert chess.Move.from_uci(best) in board.legal_moves
    assert pi.dtype == np.float32 and abs(float(pi.sum()) - 1.0) < 1e-6 and 1 <= np.count_nonzero(pi) <= 2
    legal_idx = {utils.move_to_index(m) for m in board.legal_moves}
    assert set(np.nonzero(pi)[0].tolist()) <= legal_idx


def test_run_self_play_game_on_gpu(mods):
    config, self_play, model, torch = mods["config"], mods["self_play"], mods["model"], mods["torch"]
    config.NUM_SIMULATIONS, config.MAX_GAME_MOVES = 100, 6
    np.random.seed(1)
    data = self_play.run_self_play_game(model, 3)
    assert isinstance(data, list) and len(data) == 6
    for st, pi, z in data:
        assert isinstance(st, torch.Tensor) and tuple(st.shape) == (120, 8, 8) and st.device.type == "cpu"
        assert pi.shape == (4672,) and abs(float(pi.sum()) - 1.0) < 1e-6 and z in (0.0, -0.0)
    assert float(data[0][0][98:110].sum()) == 32.0          # current position = 32 pieces in history block 7
    assert float(data[0][0][:98].sum()) == 0.0              # no history before the first move
    games = self_play.run_self_play_games(model, [10, 11, 12, 13], seeds=[1, 2, 3, 4], n_slots=4)
    assert sorted(games) == [10, 11, 12, 13] and all(len(v) == 6 for v in games.values())
