"""
config -- the constants of /root/reference/config.py that the rollout path reads (same names, same
defaults, read at CALL time so `config.NUM_SIMULATIONS = 800` before a call works as in the reference).
"""
import torch

# --- Hardware (config.py:9-10) ---
DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
USE_AMP = torch.cuda.is_available()
# Dtype regime of the evaluate stage.  False = R3 of SURVEY.md section 8 (fp32 logits -> fp32 tree
# arithmetic; the parity target).  True = what the reference does on CUDA (mcts.py:183,285:
# torch.autocast -> fp16 net).
AUTOCAST = False

# --- Chess game (config.py:13-29) ---
BOARD_SIZE = 8
INPUT_CHANNELS = 120
NUM_ACTIONS = 8 * 8 * 73

# --- MCTS (config.py:32-41) ---
NUM_SIMULATIONS = 250
CPUCT = 1.0
TEMPERATURE_INITIAL = 1.0
TEMPERATURE_FINAL = 0.1
TEMPERATURE_THRESHOLD = 30
DIRICHLET_ALPHA = 0.1
DIRICHLET_EPSILON = 0.25
WIDEN_COEFF = 1.5
MCTS_BATCH_SIZE = 96

# --- Neural network (config.py:44-48) ---
RESIDUAL_BLOCKS = 15
SE_RESIDUAL_BLOCKS = 5
CONV_FILTERS = 256
SE_REDUCTION_RATIO = 16
GRAD_CLIP_MAX = 2.0

# --- Pretraining / training (config.py:51-67; read by train.py and main.py, not by the rollout path) ---
NUM_WORKERS = 6
MID_EPOCH_CHECKPOINT = 50_000
PRETRAINING_T_MAX = 1_343_500
NUM_THREADS = 6
GAMES_MINIMUM = 100
BATCH_SIZE = 256
MAX_GAME_MOVES = 16384
LEARNING_RATE = 0.001
WEIGHT_DECAY = 1e-4
LR_MIN = 5e-7
EPOCHS_PER_ITERATION = 18
NUM_ITERATIONS = 80
CHECKPOINT_INTERVAL = 1
GAME_BUFFER_SIZE = 100000

# --- Paths (config.py:70-73) ---
PGN_DATA_DIR = "fishtest"
SAVE_DIR = "checkpoints"
LOG_DIR = "logs"
DATA_DIR = "data"

# --- engine options (new; no counterpart in the reference) ---
# "reference": BetaOne's own search semantics, bit-exact (default).  "fast": csrc/bo_fast.h -- virtual loss,
# FAST_LEAVES distinct leaves per game per step, full-width expansion; a different (conventional AlphaZero) search.
SEARCH_MODE = "reference"
FAST_LEAVES = 16
ENGINE_MAX_PLIES = 2048   # capacity of one game's position stack on the GPU
