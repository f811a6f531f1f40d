#!/usr/bin/env python3
"""scripts/nn_layout_race.py -- LAB tool (not imported by the package): time every evaluate-stage candidate (library layouts and the
hand-written kernel sets of betaone_amd/fused_net.py) on one batch shape.  The product picks its evaluate stage by shape
(betaone_amd.nn_tune.kernel_route); this race was the start-up tuner of rounds 1-2 and is kept only to re-check that rule.

    python scripts/nn_layout_race.py --net 10x128 --batch 256 --dtype fp32
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import betaone_amd  # noqa: F401
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


def timed_inference_copy(model, batch: int, device, dtype: torch.dtype = torch.float32, verbose: bool = False):
    """The start-up timing race (debugging aid): every candidate layout / kernel set is timed on the real batch shape."""
    device = torch.device(device)
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
        from betaone_amd import engine as E
        from betaone_amd.fused_net import FusedPolicyValueNet

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
        from betaone_amd import engine as E
        from betaone_amd.fused_net import FusedPolicyValueNet

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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="10x128", choices=["4x64", "10x128", "20x256"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16"])
    a = ap.parse_args()
    from betaone_amd import dropin

    dropin.install()
    import config
    import network

    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = {"4x64": (3, 1, 64), "10x128": (8, 2, 128), "20x256": (15, 5, 256)}[a.net]
    torch.manual_seed(0)
    net = network.PolicyValueNet().eval()
    best = timed_inference_copy(net, a.batch, "cuda:0", {"fp32": torch.float32, "fp16": torch.float16}[a.dtype], verbose=True)
    print("fastest:", getattr(best, "conv", None) or getattr(best, "layout", None))
