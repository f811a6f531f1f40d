"""
betaone_amd/nn_tune.py -- pick the evaluate stage's memory layout for the batch it will actually see.

The engine writes NN input rows as NCHW float32.  The evaluate stage is chosen by SHAPE (kernel_route): the hand-written gfx950
kernels wherever they exist; the library path (PyTorch-ROCm / MIOpen: NCHW at a few hundred positions per batch, channels-last
from about a thousand -- measured on MI355X, net 8+2x128 fp32: 1.48 vs 2.29 ms at 256, 4.49 vs 4.41 ms at 1024) only for shapes
without one, and then with a warning.  (The start-up timing race of rounds 1-2 is not part of the package any more: it lives on as
the lab script scripts/nn_layout_race.py.)
"""
from __future__ import annotations

import torch


def f32_pipe_default() -> bool:
    """BETAONE_F32_TOWER=fp32 keeps float32 nets on the fp32 matrix pipe (csrc/bo_tower_wg.h / bo_conv.h); the default 'split' runs
    them on the fp16 pipe with (hi, lo) operand pairs (csrc/bo_tower_s.h: same 1e-5 agreement with the float32 net, 16x the rate)."""
    import os
    v = os.environ.get("BETAONE_F32_TOWER", "split").lower()
    if v not in ("split", "fp32"):
        raise ValueError("BETAONE_F32_TOWER must be 'split' or 'fp32'")
    return v == "fp32"


def kernel_route(filters: int, batch: int, dtype: torch.dtype, f32_pipe: bool = None):
    """Which hand-written evaluate stage (betaone_amd/fused_net.py `conv=`) a net of this shape runs on -- decided by shape, not by a
    timing race -- or None where none exists (the net then stays under PyTorch-ROCm's library kernels, and says so):
      float32, 64 / 128 / 256 filters: batch <= 256 / ((filters/16)*4) (1-4 boards at 256 filters: uci.py's searches) -> 'tower_b1' (the
                                 whole tower as ONE launch of c_out/16 x 4 workgroups per board, layers handed over inside the launch);
      float32, 128 / 256 filters: batch <= 16 -> 'mfma_small' (a board's layer cut into c_out/16 x 4 workgroups, one launch per layer), else 'tower_split'
                                 (the LDS-resident tower on the fp16 matrix pipe, float32 operands as (hi, lo) fp16 pairs);
                                 with f32_pipe (or BETAONE_F32_TOWER=fp32): 'tower_wg' (fp32-MFMA Winograd tower, 128 filters) /
                                 'mfma' (per-layer implicit GEMM on the fp32 pipe, 256 filters);
      float32, 64 filters:       batch <= 16 -> 'mfma_small', else 'tower_wg';
      float16, 128 / 256 filters: 'tower_f16' (csrc/bo_tower_h.h, two boards per workgroup)."""
    if f32_pipe is None:
        f32_pipe = f32_pipe_default()
    if dtype == torch.float32:
        if filters in (64, 128, 256) and batch * (filters // 16) * 4 <= 256:
            return "tower_b1"
        if filters in (64, 128, 256) and batch <= 16:
            return "mfma_small"
        if filters in (128, 256) and not f32_pipe:
            return "tower_split"
        if filters in (64, 128):
            return "tower_wg"
        if filters == 256:
            return "mfma"
    if dtype == torch.float16 and filters in (128, 256):
        return "tower_f16"
    return None


def best_inference_copy(model, batch: int, device, dtype: torch.dtype = torch.float32, verbose: bool = False, f32_pipe=None):
    """BN-folded inference copy of a PolicyValueNet for `batch` rows: the hand-written kernels chosen by kernel_route(); where
    there are none (other filter counts, bfloat16) the PyTorch-ROCm copy in NCHW up to 512 rows and channels-last beyond, with
    a warning naming the library path."""
    import warnings

    device = torch.device(device)
    if not hasattr(model, "for_inference"):
        return model
    if device.type != "cuda":
        return model.for_inference(dtype=dtype, channels_last=False)
    from . import engine as E
    from .fused_net import FusedPolicyValueNet

    filters = model.conv_input.out_channels
    route = kernel_route(filters, batch, dtype, f32_pipe)
    if route is not None:
        try:
            net = FusedPolicyValueNet(model.to(device), conv=route, f32_pipe=f32_pipe).to(device)
            net.route = route
            if verbose:
                print(f"[nn_route] batch={batch} filters={filters} {dtype}: hand-written evaluate stage conv='{route}'")
            return net
        except E.EngineError as ex:  # (e.g. an SE block wider than the tower kernels take)
            why = str(ex)
    else:
        why = f"no hand-written evaluate stage for {filters} filters in {dtype}"
    cl = batch > 512
    net = model.to(device).for_inference(dtype=dtype, channels_last=cl)
    net.layout = "channels_last" if cl else "nchw"
    net.route = "pytorch-rocm library kernels"
    warnings.warn(f"betaone_amd: the evaluate stage of this net runs on PyTorch-ROCm library kernels (MIOpen / hipBLASLt), {net.layout}: {why}",
                  RuntimeWarning, stacklevel=2)
    return net
