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
# Knob: set GPU_MAX_HW_QUEUES yourself before importing to override (documented in INTEGRATION.md); hw_queues() reports what is
# in effect and whether it can still take effect.
_HWQ_PRESET = _os.environ.get("GPU_MAX_HW_QUEUES")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def hw_queues() -> dict:
    """{'value': GPU_MAX_HW_QUEUES in this process's environment, 'source': 'caller' | 'betaone_amd default',
    'in_effect': False if the HIP runtime had already been initialised when this package was imported (the variable is read
    once, at runtime start-up: a later setting changes nothing)}."""
    return {"value": int(_os.environ.get("GPU_MAX_HW_QUEUES", "4")), "source": "caller" if _HWQ_PRESET is not None else "betaone_amd default",
            "in_effect": not _HIP_WAS_UP}


def _hip_already_up() -> bool:
    import sys

    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


_HIP_WAS_UP = _hip_already_up()
if _HIP_WAS_UP and _HWQ_PRESET is None:
    import warnings as _w

    _w.warn("betaone_amd imported after the HIP runtime was initialised: GPU_MAX_HW_QUEUES=8 cannot take effect any more "
            "(the runtime keeps its default of 4 hardware queues; the finished-game hand-over then queues behind the search "
            "as soon as an RCCL communicator exists, ~0.2 ms per ply).  Import betaone_amd before the first torch.cuda call "
            "or export GPU_MAX_HW_QUEUES yourself.", RuntimeWarning, stacklevel=2)

__version__ = "0.1.0"
