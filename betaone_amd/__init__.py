"""
betaone_amd -- MI355X-native self-play rollout engine behind BetaOne's own Python surface.

Only the hot path of kevinh-e/BetaOne is here: the MCTS + NN self-play loop
(`run_self_play_game` / `run_mcts` / `PolicyValueNet`, reference self_play.py / mcts.py /
network.py / utils.py).  The tree lives in HBM and is advanced by hand-written gfx950 kernels
(betaone_amd/csrc, C ABI in include/betaone_engine.h); the policy/value net runs under
PyTorch-ROCm.  There is no CPU fallback: importing the engine without the HIP library raises.
"""
__version__ = "0.1.0"
