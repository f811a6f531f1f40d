"""
betaone_amd.dropin -- BetaOne's own module names over the MI355X engine.

The reference is a flat directory whose callers do `import config`, `from network import
PolicyValueNet`, `from mcts import run_mcts`, `from self_play import run_self_play_game,
save_game_data`, `import utils` (main.py:18-23, uci.py:15-19, train.py:26-29).  Putting THIS
directory first on sys.path makes those imports resolve to the engine-backed modules, so main.py,
train.py and uci.py run unchanged.
"""
import os
import sys

DIR = os.path.dirname(os.path.abspath(__file__))


def install() -> str:
    """Put the drop-in modules first on sys.path (idempotent) and return the directory."""
    if DIR in sys.path:
        sys.path.remove(DIR)
    sys.path.insert(0, DIR)
    return DIR
