#!/usr/bin/env python3
"""tests/uci_latency.py -- BASELINE.json configs[3]: single-position analysis through the drop-in run_mcts
(the call uci.py makes, uci.py:63,84): 1600 sims/move, 20-block x 256 net, one MI355X, hipGraph on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle", "shim")]
import numpy as np, torch
from betaone_amd import dropin
dropin.install()
import chess, config, mcts, network, utils   # chess = oracle/shim stand-in for python-chess (host-side objects only)

sims = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
config.NUM_SIMULATIONS = sims
config.POLICY_SOFTMAX = os.environ.get('BO_UCI_SOFTMAX', config.POLICY_SOFTMAX)
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
torch.manual_seed(0)
model = network.PolicyValueNet().to("cuda").eval()
board = chess.Board()
tracker = utils.RepetitionTracker(); tracker.add_board(board)
history = [board.copy()]
for u in "e2e4 e7e5 g1f3 b8c6 f1b5 a7a6".split():
    board.push(chess.Move.from_uci(u)); tracker.add_board(board); history.append(board.copy())
hist = history[-8:][-7:]
np.random.seed(0)
for _ in range(3):
    mcts.run_mcts(board, model, hist, tracker)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); best, pi = mcts.run_mcts(board, model, hist, tracker); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
evals = 1 + -(-sims // config.MCTS_BATCH_SIZE)
print(f"run_mcts {sims} sims, net 15+5x256 fp32, batch-1: median {np.median(ts):.2f} ms  min {ts.min():.2f} ms  "
      f"({evals} NN evaluations, {np.median(ts)/evals:.3f} ms per evaluation+step)  best={best.uci()}  "
      f"nodes/s={sims/np.median(ts)*1e3:.0f}")
