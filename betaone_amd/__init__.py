"""
betaone_amd -- MI355X-native self-play rollout engine behind BetaOne's own Python surface.

Only the hot path of kevinh-e/BetaOne is here: the MCTS + NN self-play loop
(`run_self_play_game` / `run_mcts` / `PolicyValueNet`, reference self_play.py / mcts.py /
network.py / utils.py).  The tree lives in HBM and is advanced by hand-written gfx950 kernels
(betaone_amd/csrc, C ABI in include/betaone_engine.h); the policy/value net runs under
PyTorch-ROCm.  There is no CPU fallback: importing the engine without the HIP library raises.
"""
import os as _os

# The engine drives a handful of HIP streams at once: the search, finished games' hand-over beside it, the record exchange
# and RCCL's own.  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one queue run in
# order, and with the hand-over queued behind a whole ply of search every rank lost 0.2-0.25 ms per ply as soon as an RCCL
# communicator added its streams (measured; 2 queues cost the single-process run 0.13 ms).  Read when the HIP runtime
# starts, so it has to be in the environment before the first GPU call: importing this package first is enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
