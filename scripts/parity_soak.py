#!/usr/bin/env python3
"""scripts/parity_soak.py [scale] -- one-off differential soak of the GPU engine against the CPU oracle (TEST INFRASTRUCTURE,
not part of the product): random-position move generation, many concurrent searches, whole games to termination; all
bit-exact or it raises.  scale 1 takes ~15 s on an MI355X box, scale 7 (4 000 random games, 400 x 30-ply games, 120 full
games) ~90 s -- run clean at the end of round 1."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch  # noqa: F401  (loads PyTorch-ROCm's HIP runtime first)
from betaone_amd import engine as E
E.load_hip_library()
import engine_cases as EC

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t = time.time()
EC.check_movegen_random_positions("hip", n_games=600 * k, max_plies=220, seed=12345 + k)
print("move generation on random positions == oracle", round(time.time() - t, 1), "s", flush=True)
t = time.time()
EC.check_multi_game_vs_oracle("hip", n_games=64 * k, plies=14 + 2 * k, sims=96, batch=16)
print("concurrent self-play prefixes == oracle", round(time.time() - t, 1), "s", flush=True)
t = time.time()
EC.check_full_games_vs_oracle("hip", n_games=16 * k, sims=16, batch=8, max_game_moves=600)
print("full games to termination == oracle", round(time.time() - t, 1), "s", flush=True)
