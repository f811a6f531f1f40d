"""
betaone_amd/nn_tune.py -- pick the evaluate stage's memory layout for the batch it will actually see.

The engine writes NN input rows as NCHW float32.  Under PyTorch-ROCm/MIOpen the residual tower is faster in
plain NCHW at a few hundred positions per batch and faster in channels-last from about a thousand
(measured on MI355X, net 8+2x128 fp32: 1.48 vs 2.29 ms at 256, 4.49 vs 4.41 ms at 1024), so the layout is
chosen by timing both once on the real batch shape.
"""
from __future__ import annotations

import time

import torch


def _time_forward(net, x, reps: int = 8) -> float:
    """Seconds per forward: best of three timed rounds after at least 50 ms of warm-up (the first launches after the
    host-side weight packing of a candidate run at idle clocks and would misrank it)."""
    with torch.no_grad():
        t0 = time.perf_counter()
        n = 0
        while n < 3 or time.perf_counter() - t0 < 0.05:
            net(x)
            torch.cuda.synchronize(x.device)
            n += 1
        replay = None
        try:  # time what the rollout runs: a captured hipGraph of the forward (eager timing at batch 1 is launch-bound)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                net(x)
            replay = g.replay
        except Exception:
            torch.cuda.synchronize(x.device)
        run = replay if replay is not None else (lambda: net(x))
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize(x.device)
            t = (time.perf_counter() - t0) / reps
            best = t if best is None or t < best else best
    return best


def best_inference_copy(model, batch: int, device, dtype: torch.dtype = torch.float32, verbose: bool = False):
    """BN-folded inference copy of a PolicyValueNet in whichever layout runs faster for `batch` rows."""
    device = torch.device(device)
    if not hasattr(model, "for_inference"):
        return model
    if device.type != "cuda":
        return model.for_inference(dtype=dtype, channels_last=False)
    x = torch.zeros((batch, 120, 8, 8), dtype=dtype, device=device)
    best, best_t, best_cl = None, None, None
    for cl in (False, True):
        net = model.to(device).for_inference(dtype=dtype, channels_last=cl)
        t = _time_forward(net, x)
        if verbose:
            print(f"[nn_tune] batch={batch} channels_last={cl}: {t * 1e3:.3f} ms")
        if best_t is None or t < best_t:
            best, best_t, best_cl = net, t, cl
    best.layout = "channels_last" if best_cl else "nchw"
    if dtype == torch.float32:  # NCHW fp32 with the hand-written kernels (csrc/bo_nn_fused.h, csrc/bo_conv.h)
        from . import engine as E
        from .fused_net import FusedPolicyValueNet

        for conv in (("miopen", "mfma_small") if batch <= 16 else ("miopen", "mfma", "tower", "tower_wg")):
            try:
                fused = FusedPolicyValueNet(model.to(device), conv=conv).to(device)
            except E.EngineError:
                if conv != "miopen":  # filter count without an MFMA instantiation
                    continue
                raise
            t = _time_forward(fused, x)
            if verbose:
                print(f"[nn_tune] batch={batch} nchw fused epilogues, conv={conv}: {t * 1e3:.3f} ms")
            if t < best_t:
                best, best_t = fused, t
    if dtype == torch.float16:  # fp16 tower, two boards per workgroup (csrc/bo_tower_h.h); it takes the float32 planes itself
        from . import engine as E
        from .fused_net import FusedPolicyValueNet

        try:
            fused = FusedPolicyValueNet(model.to(device), conv="tower_f16").to(device)
            t = _time_forward(fused, x)
            if verbose:
                print(f"[nn_tune] batch={batch} fp16 tower: {t * 1e3:.3f} ms")
            if t < best_t:
                best = fused
        except E.EngineError:
            pass
    return best
