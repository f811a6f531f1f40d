"""
config -- the module-level constants of /root/reference/config.py, same names and default values, read at CALL
time by the rollout path (so `config.NUM_SIMULATIONS = 800` before a call works exactly as in the reference).
Constants are grouped in tables and published as module attributes.
"""
import os

import torch

# reference config.py:9 probes torch.cuda.is_available() at import; BETAONE_DEVICE (e.g. "cpu" for host-side helper
# processes that must not open the GPU, or "cuda:3") states the device without probing
_DEVICE_ENV = os.environ.get("BETAONE_DEVICE", "")
_HAS_GPU = _DEVICE_ENV.startswith("cuda") if _DEVICE_ENV else torch.cuda.is_available()

_TABLES = {
    # reference config.py:9-10
    "hardware": dict(DEVICE=_DEVICE_ENV or ("cuda" if _HAS_GPU else "cpu"), USE_AMP=_HAS_GPU),
    # config.py:13-29 -- 8 history blocks x (12 piece + 2 repetition planes) + 8 scalar planes; 8x8x73 move planes
    "encoding": dict(BOARD_SIZE=8, INPUT_CHANNELS=8 * 14 + 8, NUM_ACTIONS=64 * 73),
    # config.py:32-41
    "search": dict(NUM_SIMULATIONS=250, CPUCT=1.0, TEMPERATURE_INITIAL=1.0, TEMPERATURE_FINAL=0.1,
                   TEMPERATURE_THRESHOLD=30, DIRICHLET_ALPHA=0.1, DIRICHLET_EPSILON=0.25, WIDEN_COEFF=1.5,
                   MCTS_BATCH_SIZE=96),
    # config.py:44-48
    "network": dict(RESIDUAL_BLOCKS=15, SE_RESIDUAL_BLOCKS=5, CONV_FILTERS=256, SE_REDUCTION_RATIO=16, GRAD_CLIP_MAX=2.0),
    # config.py:51-67 -- read by train.py / main.py, not by the rollout path (MAX_GAME_MOVES is, self_play.py:102)
    "training": dict(NUM_WORKERS=6, MID_EPOCH_CHECKPOINT=50_000, PRETRAINING_T_MAX=1_343_500, NUM_THREADS=6,
                     GAMES_MINIMUM=100, BATCH_SIZE=256, MAX_GAME_MOVES=16384, LEARNING_RATE=0.001, WEIGHT_DECAY=1e-4,
                     LR_MIN=5e-7, EPOCHS_PER_ITERATION=18, NUM_ITERATIONS=80, CHECKPOINT_INTERVAL=1,
                     GAME_BUFFER_SIZE=100000),
    # config.py:70-73
    "paths": dict(PGN_DATA_DIR="fishtest", SAVE_DIR="checkpoints", LOG_DIR="logs", DATA_DIR="data"),
    # new, no counterpart in the reference:
    #   AUTOCAST      False = dtype regime R3 of SURVEY.md section 8 (fp32 net -> fp32 tree, the parity target);
    #                 True  = what the reference does on CUDA (torch.autocast around the net, mcts.py:183,285)
    #   SEARCH_MODE   "reference" = BetaOne's own search semantics, bit-exact; "fast" = csrc/bo_fast.h (virtual loss,
    #                 FAST_LEAVES leaves per game per step, full-width expansion: a conventional AlphaZero search)
    #   ENGINE_MAX_PLIES  capacity of one game's position stack on the GPU; None = MAX_GAME_MOVES + 2 in self-play
    #                 (2.7 MB per slot at 16384; a smaller value stops longer games like the move limit does)
    #   POLICY_SOFTMAX  "torch" = torch.softmax(logits, dim=1) exactly where the reference calls it (mcts.py:185,287), inside
    #                 the captured graph; the engine gathers probabilities (the seam the parity tests record).
    #                 "engine" = the step kernel's own softmax over the logits row (hardware exp; within 1e-5 relative)
    #   COHORTS       run_self_play_games: the resident games as UP TO this many phase-shifted cohorts, each on its own HIP stream
    #                 (rollout.CohortRollout: one cohort's tower overlaps another's tree step, heads and ply boundary); 1 = one Rollout.
    #                 Results do not depend on it (games are independent).  The largest of COHORTS, COHORTS/2, ... that divides the slot
    #                 count and leaves every cohort at least 64 slots is used (below that a cohort's kernels no longer fill their share of
    #                 the CUs).  From three cohorts up every cohort's stream is confined to its own share of the compute units
    #                 (BETAONE_COHORT_CU_MASK=off/contiguous/interleaved overrides).
    "engine": dict(AUTOCAST=False, SEARCH_MODE="reference", FAST_LEAVES=16, ENGINE_MAX_PLIES=None, POLICY_SOFTMAX="torch", COHORTS=4),
}
for _group in _TABLES.values():
    globals().update(_group)
del _group
