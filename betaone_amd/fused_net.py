"""
betaone_amd/fused_net.py -- the evaluate stage with hand-written fused epilogues.

Same function as PolicyValueNet.forward (/root/reference/network.py:167-198) in eval mode: BatchNorm is folded into
the convolutions and every conv is followed by (or fused with) a gfx950 epilogue instead of the 3-8 separate
elementwise launches PyTorch issues (bias, ReLU, residual add, SE pooling / FCs / sigmoid / scale).  Two variants:

  conv="miopen": the 3x3 convolutions stay in MIOpen (fp32 Winograd asm kernel under PyTorch-ROCm), each followed
                 by ONE kernel from csrc/bo_nn_fused.h;
  conv="mfma":   the 3x3 convolutions run in csrc/bo_conv.h (direct implicit GEMM on the fp32 matrix cores, bias /
                 ReLU / residual fused into its epilogue); SE blocks add csrc/bo_nn_fused.h's SE kernel.

  conv="mfma_small": the small-batch form of "mfma" (bo_k_conv3x3_small: a board's layer is spread over (c_out/16) x 4
                 workgroups) for uci.py's single-position searches (BASELINE.json configs[3]).
  conv="tower_b1": the whole tower of ONE board (up to 256 / ((filters/16)*4) boards) as ONE launch of the same (c_out/16) x 4 workgroups
                 per board, the layers handed over inside the launch (csrc/bo_tower_b1.h) instead of one launch per layer: uci.py's
                 single-position searches (BASELINE.json configs[3]); larger batches fall back to "mfma_small" launches.
  conv="tower":  input conv + all residual blocks are ONE persistent kernel that keeps each board's activations in
                 LDS (csrc/bo_tower.h); 64 or 128 filters.
  conv="tower_wg": the same with Winograd F(2x2,3x3) convolutions (csrc/bo_tower_wg.h), 2.25x fewer MFMA cycles.
  conv="tower_split": float32 in and out on the fp16 matrix pipe (csrc/bo_tower_s.h): weights and activations as (hi, lo)
                 fp16 pairs, three MFMAs per product, float32 accumulation; 128 or 256 filters, one board per workgroup.
  conv="tower_f16": fp16 weights/activations, fp32 accumulation (csrc/bo_tower_h.h), 128 or 256 filters, two boards
                 per workgroup; the evaluate stage of BASELINE.json configs[4].  Takes the engine's float32 planes.

NCHW float32 only; other dtypes/layouts use PolicyValueNet.for_inference().
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

import ctypes as C
import os

import numpy as np

from . import engine as E


MFMA_CONV_SHAPES = {(120, 64), (64, 64), (120, 128), (128, 128), (120, 256), (256, 256)}


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """[c_out, c_in, 3, 3] -> [tap 9][c_in/8][c_out][k 2][e 4], element = W[oc][8*t4 + 2*e + k][tap]: the A fragments
    of four consecutive 32x32x2 MFMA K-steps as one 16-byte load per lane (csrc/bo_conv.h)."""
    co, ci = w.shape[0], w.shape[1]
    return w.reshape(co, ci // 8, 4, 2, 9).permute(4, 1, 0, 3, 2).contiguous()


_WG_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)


def winograd_k_order(c_out: int, c_in: int) -> list:
    """Input channel of (K-step, row k) of a BO_TOWER_WINOGRAD layer, flattened [step][k] (include/betaone_engine.h).
    128 filters: K-step 4c + sl covers channels 16*(4*(c & 1) + sl) + 4*(c >> 1) + k -- the four channels that wave
    4*(c & 1) + sl of the kernel produced itself in the previous layer; otherwise the natural order 4*step + k."""
    if c_out == 128 and c_in == 128:
        return [16 * (4 * (c & 1) + sl) + 4 * (c >> 1) + k for c in range(8) for sl in range(4) for k in range(4)]
    return list(range(c_in))


def pack_conv_weight_winograd(w: torch.Tensor) -> torch.Tensor:
    """[c_out, c_in, 3, 3] -> U = G g G^T laid out [c_in/4][c_out/16][4][64][4] (include/betaone_engine.h,
    BO_TOWER_WINOGRAD): element (step, ob, pq, lane, e) = U[4*pq + e][16*ob + (lane & 15)][channel(step, lane >> 4)],
    channel() = winograd_k_order."""
    co, ci = w.shape[0], w.shape[1]
    u = torch.einsum("ai,ocij,bj->aboc", _WG_G, w.double(), _WG_G).reshape(16, co, ci)  # [pos][oc][ic]
    u = u[:, :, winograd_k_order(co, ci)]
    u = u.reshape(4, 4, co // 16, 16, ci // 4, 4)  # [pq][e][ob][o16][step][k]
    return u.permute(4, 2, 0, 5, 3, 1).contiguous().float()  # [step][ob][pq][k][o16][e]


def pack_conv_weight_f16(w: torch.Tensor) -> torch.Tensor:
    """[c_out, c_in, 3, 3] -> fp16 [9*c_in/16][c_out/32][64][8] (include/betaone_engine.h, BO_TOWER_DIRECT_F16):
    element (step, mt, lane, i) = W[32*mt + (lane & 31)][16*(step % (c_in/16)) + 8*(lane >> 5) + i][tap = step / (c_in/16)]."""
    co, ci = w.shape[0], w.shape[1]
    u = w.reshape(co // 32, 32, ci // 16, 2, 8, 9)  # [mt][o][cg][kg][i][tap]
    return u.permute(5, 2, 0, 3, 1, 4).contiguous().half()  # [tap][cg][mt][kg][o][i]


def pack_conv_weight_f16_t16(w: torch.Tensor) -> torch.Tensor:
    """[c_out, c_in, 3, 3] -> fp16 [9][c_in/32][c_out/16][64][8] (BO_TOWER_DIRECT_F16_T16: the A fragments of v_mfma_f32_16x16x32_f16):
    element (tap, g, ot, lane, i) = W[16*ot + (lane & 15)][32*g + 8*(lane >> 4) + i][tap]."""
    co, ci = w.shape[0], w.shape[1]
    u = w.reshape(co // 16, 16, ci // 32, 4, 8, 9)  # [ot][o][g][kb][i][tap]
    return u.permute(5, 2, 0, 3, 1, 4).contiguous().half()  # [tap][g][ot][kb][o][i]


# which MFMA tile conv='tower_f16' multiplies with, by filter count ("16": csrc/bo_tower_h16.h); BETAONE_F16_TILE overrides.  Same-box A/Bs
# (profiles/r05_all_configs.md): 128 filters (fast mode) 4.78 s against 5.00 s per ply of 16 384 games with 16x16x32 tiles; 256 filters
# (configs[4]) 13.0 against 12.2-12.5 ms -- the 256-filter instance of the new tiling spills registers.
F16_TILE_DEFAULT = {128: "16", 256: "32"}


def split_scale(w: torch.Tensor) -> float:
    """The power of two s that puts the largest |s*w| in [2^14, 2^15) (BO_TOWER_SPLIT_F16: fp16 pairs of s*w)."""
    m = float(w.abs().max())
    if m == 0.0 or not np.isfinite(m):
        return 1.0
    return float(2.0 ** (14 - int(np.floor(np.log2(m)))))


def split_f16(w: torch.Tensor):
    """(hi, lo) fp16 with hi = RN16(w), lo = RN16(w - hi), computed in float64."""
    w = w.double()
    hi = w.half()
    lo = (w - hi.double()).half()
    return hi, lo


def pack_conv_weight_split(w: torch.Tensor, scale: float) -> torch.Tensor:
    """[c_out, c_in, 3, 3] float32 -> fp16 [9*c_in/16][c_out/32][2 = hi, lo][64][8] of scale*w (include/betaone_engine.h,
    BO_TOWER_SPLIT_F16; the element order of pack_conv_weight_f16 with every fragment doubled)."""
    co, ci = w.shape[0], w.shape[1]
    hi, lo = split_f16(w.double() * scale)
    u = torch.stack([hi, lo], 0).reshape(2, co // 32, 32, ci // 16, 2, 8, 9)  # [hl][mt][o][cg][kg][i][tap]
    return u.permute(6, 3, 1, 0, 4, 2, 5).contiguous()  # [tap][cg][mt][hl][kg][o][i]


SPLIT_TILE_DEFAULT = "16"  # which MFMA tile conv='tower_split' multiplies with at 128 filters ("16": csrc/bo_tower_s16.h); BETAONE_SPLIT_TILE overrides


def pack_conv_weight_split16(w: torch.Tensor, scale: float) -> torch.Tensor:
    """[c_out, c_in, 3, 3] float32 -> fp16 [9][c_in/32][c_out/16][2 = hi, lo][64][8] of scale*w (BO_TOWER_SPLIT_F16_T16: the A fragments of
    v_mfma_f32_16x16x32_f16): element (tap, g, ot, hl, lane, i) = split(scale * W[16*ot + (lane & 15)][32*g + 8*(lane >> 4) + i][tap])."""
    co, ci = w.shape[0], w.shape[1]
    hi, lo = split_f16(w.double() * scale)
    u = torch.stack([hi, lo], 0).reshape(2, co // 16, 16, ci // 32, 4, 8, 9)  # [hl][ot][o][g][kb][i][tap]
    return u.permute(6, 3, 1, 0, 4, 2, 5).contiguous()  # [tap][g][ot][hl][kb][o][i]


def pack_conv_weight_small(w: torch.Tensor) -> torch.Tensor:
    """[c_out, c_in, 3, 3] -> [c_out/16][tap 9][c_in/16][64][4] (bo_nn_conv3x3_small): element (ot, tap, g, lane, e) =
    W[16*ot + (lane & 15)][16*g + 4*e + (lane >> 4)][tap]."""
    co, ci = w.shape[0], w.shape[1]
    u = w.reshape(co // 16, 16, ci // 16, 4, 4, 9)  # [ot][o][g][e][k][tap]
    return u.permute(0, 5, 2, 4, 1, 3).contiguous()  # [ot][tap][g][k][o][e]


def pack_conv_weight_small_split(w: torch.Tensor, scale: float) -> torch.Tensor:
    """[c_out, c_in, 3, 3] float32 -> fp16 [c_out/16][tap 9][c_in/16][64][2 = hi, lo][4] of scale*w (bo_nn_b1_create, weights_split_dev):
    element (ot, tap, g, lane, hl, j) = split(scale * W[16*ot + (lane & 15)][16*g + 4*(lane >> 4) + j][tap]) -- per lane the B fragment
    of one v_mfma_f32_16x16x16_f16 K-step as (hi x4 | lo x4) = 16 bytes, at the index the float32 fragments of pack_conv_weight_small have."""
    co, ci = w.shape[0], w.shape[1]
    hi, lo = split_f16(w.double() * scale)
    u = torch.stack([hi, lo], 0).reshape(2, co // 16, 16, ci // 16, 4, 4, 9)  # [hl][ot][n][g][kq][j][tap]
    return u.permute(1, 6, 3, 4, 2, 0, 5).contiguous()  # [ot][tap][g][kq][n][hl][j]


def conv3x3_small(lib, x, wpacked, bias, c_in, c_out, mode=0, residual=None):
    """y = epilogue(conv3x3(x)) through bo_nn_conv3x3_small; x NCHW float32 [B, c_in_x <= c_in, 8, 8], contiguous."""
    B = x.shape[0]
    y = torch.empty((B, c_out, 8, 8), dtype=torch.float32, device=x.device)
    rc = lib.bo_nn_conv3x3_small(x.data_ptr(), wpacked.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                                 y.data_ptr(), B, c_in, x.shape[1], c_out, mode, torch.cuda.current_stream(x.device).cuda_stream)
    if rc:
        raise E.EngineError(lib.bo_last_error().decode())
    return y


def conv3x3_mfma(lib, x, wpacked, bias, c_out, mode=0, residual=None, out=None):
    """y = epilogue(conv3x3(x)) through the C ABI (bo_nn_conv3x3); x NCHW float32 [B, c_in, 8, 8], contiguous."""
    B, c_in = x.shape[0], x.shape[1]
    if not x.is_contiguous() or x.dtype != torch.float32 or x.shape[2:] != (8, 8):
        raise E.EngineError("conv3x3_mfma: x must be a contiguous float32 [B, C, 8, 8] tensor")
    y = out if out is not None else torch.empty((B, c_out, 8, 8), dtype=torch.float32, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.bo_nn_conv3x3(x.data_ptr(), wpacked.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                           y.data_ptr(), B, c_in, c_out, mode, stream)
    if rc:
        raise E.EngineError(lib.bo_last_error().decode())
    return y


class FusedPolicyValueNet(nn.Module):
    def __init__(self, net, conv="miopen", f32_pipe=None):
        """net: a PolicyValueNet (any device); weights are copied, BN folded.  f32_pipe (conv='tower_b1' only; None = the
        BETAONE_F32_TOWER setting): True keeps the tiles on the fp32 matrix pipe (exact float32 products), False multiplies them on
        the fp16 pipe with (hi, lo) operand pairs -- the precision of conv='tower_split'."""
        super().__init__()
        self.lib = E.load_hip_library()
        self.conv = conv
        self._f32_pipe = f32_pipe
        # policy FC + softmax + value head as one kernel behind the Winograd tower (needs contiguous float32 Linear weights of the
        # reference's head shapes: 2 policy planes, 32 value planes, 256 hidden units)
        self.fused_heads = conv in ("tower_wg", "tower", "mfma", "mfma_small", "tower_b1", "tower_f16", "tower_split")
        f = net.for_inference(dtype=torch.float32, channels_last=False)
        dev = next(f.parameters()).device
        if dev.type != "cuda":
            raise E.EngineError("FusedPolicyValueNet needs the net on an MI355X (cuda device)")

        def cw(conv):
            return nn.Parameter(conv.weight.detach().contiguous(), requires_grad=False), \
                nn.Parameter(conv.bias.detach().contiguous(), requires_grad=False)

        self.w_in, self.b_in = cw(f.conv_input)
        self.blocks = []
        for i, blk in enumerate(f.residual_tower):
            w1, b1 = cw(blk.conv1)
            w2, b2 = cw(blk.conv2)
            se = None
            if hasattr(blk, "seblock"):
                se = (nn.Parameter(blk.seblock.excitation[0].weight.detach().contiguous(), requires_grad=False),
                      nn.Parameter(blk.seblock.excitation[2].weight.detach().contiguous(), requires_grad=False))
            for k, p in (("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2)):
                self.register_parameter(f"blk{i}_{k}", p)
            if se:
                self.register_parameter(f"blk{i}_se1", se[0])
                self.register_parameter(f"blk{i}_se2", se[1])
            self.blocks.append((w1, b1, w2, b2, se))
        # the two 1x1 head convolutions share their input: one conv with 2 + 32 output channels
        self.w_head = nn.Parameter(torch.cat([f.policy_conv.weight, f.value_conv.weight], 0).detach().contiguous(), requires_grad=False)
        self.b_head = nn.Parameter(torch.cat([f.policy_conv.bias, f.value_conv.bias], 0).detach().contiguous(), requires_grad=False)
        self.n_policy_ch = f.policy_conv.out_channels
        self.policy_fc, self.value_fc1, self.value_fc2 = f.policy_fc, f.value_fc1, f.value_fc2
        self.layout = "nchw+fused"
        if conv == "mfma":
            shapes = {(self.w_in.shape[1], self.w_in.shape[0])} | {(b[0].shape[1], b[0].shape[0]) for b in self.blocks}
            if not shapes <= MFMA_CONV_SHAPES:
                raise E.EngineError(f"conv='mfma' supports (c_in, c_out) in {sorted(MFMA_CONV_SHAPES)}, got {sorted(shapes)}")
            self.c = self.w_in.shape[0]
            self.p_in = nn.Parameter(pack_conv_weight(self.w_in), requires_grad=False)
            self.packed = []
            for i, (w1, _, w2, _, _) in enumerate(self.blocks):
                p1 = nn.Parameter(pack_conv_weight(w1), requires_grad=False)
                p2 = nn.Parameter(pack_conv_weight(w2), requires_grad=False)
                self.register_parameter(f"blk{i}_p1", p1)
                self.register_parameter(f"blk{i}_p2", p2)
                self.packed.append((p1, p2))
            self.zero_bias = nn.Parameter(torch.zeros(self.c, device=dev), requires_grad=False)
            self.layout = "nchw+mfma"
        elif conv in ("mfma_small", "tower_b1"):
            c = self.w_in.shape[0]
            if c not in (64, 128, 256) or self.w_in.shape[1] != 120:
                raise E.EngineError(f"conv='{conv}' supports 120 input planes and 64, 128 or 256 filters")
            self.c = c
            w0 = torch.zeros((c, 128, 3, 3), device=dev)
            w0[:, :120] = self.w_in
            self.p_in = nn.Parameter(pack_conv_weight_small(w0), requires_grad=False)
            self.packed = []
            for i, (w1, _, w2, _, _) in enumerate(self.blocks):
                p1 = nn.Parameter(pack_conv_weight_small(w1), requires_grad=False)
                p2 = nn.Parameter(pack_conv_weight_small(w2), requires_grad=False)
                self.register_parameter(f"blk{i}_p1", p1)
                self.register_parameter(f"blk{i}_p2", p2)
                self.packed.append((p1, p2))
            self.zero_bias = nn.Parameter(torch.zeros(c, device=dev), requires_grad=False)
            self.layout = "nchw+" + conv
            if conv == "tower_b1":
                self._build_b1(dev)
        elif conv in ("tower", "tower_wg"):
            self._build_tower(dev, winograd=conv == "tower_wg")
            self.layout = "nchw+" + conv
        elif conv == "tower_f16":
            self._build_tower_f16(dev)
            self.layout = "nchw+tower_f16"
        elif conv == "tower_split":
            self._build_tower_split(dev)
            self.layout = "nchw+tower_split"
        elif conv != "miopen":
            raise ValueError("conv must be 'miopen', 'mfma', 'mfma_small', 'tower_b1', 'tower', 'tower_wg', 'tower_split' or 'tower_f16'")

    def _build_b1(self, dev):
        """bo_nn_b1_create over the SAME device tensors the per-layer route uses (pack_conv_weight_small layout): the handle keeps
        their addresses, the module keeps them alive.  Unless the fp32 matrix pipe is asked for, every layer also gets (hi, lo) fp16
        weights (pack_conv_weight_small_split, scaled per layer by a power of two like conv='tower_split')."""
        from .nn_tune import f32_pipe_default

        c = self.c
        split = not (f32_pipe_default() if self._f32_pipe is None else self._f32_pipe)
        self.b1_precision = "f16 pairs (3 MFMAs per product, f32 accumulate)" if split else "f32 MFMA"
        descs, self._b1_split = [], []

        def desc(w, wp, bias, c_in, c_in_x, mode, se=None):
            ws, inv = None, 0.0
            if split:
                wf = w.detach().float().cpu()
                sc = split_scale(wf)
                ws = nn.Parameter(pack_conv_weight_small_split(wf, sc).to(dev), requires_grad=False)
                self._b1_split.append(ws)
                inv = 1.0 / sc
            d = E.BoB1LayerDesc(wp.data_ptr(), bias.data_ptr(), se[0].data_ptr() if se else None, se[1].data_ptr() if se else None,
                                c_in, c_in_x, mode, se[0].shape[0] if se else 0, ws.data_ptr() if ws is not None else None, inv, 0)
            descs.append(d)

        w0 = torch.zeros((c, 128, 3, 3))
        w0[:, :120] = self.w_in.detach().float().cpu()
        desc(w0, self.p_in, self.b_in, 128, 120, 0)
        for (w1, b1, w2, b2, se), (p1, p2) in zip(self.blocks, self.packed):
            if se is not None and se[0].shape[0] > 16:
                raise E.EngineError("conv='tower_b1' supports SE hidden widths up to 16")
            desc(w1, p1, b1, c, c, 0)
            desc(w2, p2, b2, c, c, 2 if se is not None else 1, se)
        arr = (E.BoB1LayerDesc * len(descs))(*descs)
        self._b1_max = 256 // ((c // 16) * 4)
        handle = C.c_void_p()
        rc = self.lib.bo_nn_b1_create(arr, len(descs), c, self._b1_max, dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(handle))
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        self._b1, self._b1_dev = handle, dev

    def _tower_b1(self, x):
        """Tower output [B, C, 8, 8] from ONE launch (bo_nn_b1_forward); batches beyond the resident grid take the per-layer launches."""
        B = x.shape[0]
        if B > self._b1_max:
            return self._tower_small(x)
        if x.device != self._b1_dev or x.dtype != torch.float32 or x.shape[1:] != (120, 8, 8):
            raise E.EngineError("tower_b1: x must be float32 [B, 120, 8, 8] on the net's device")
        x = x.contiguous()
        y = torch.empty((B, self.c, 8, 8), dtype=torch.float32, device=x.device)
        rc = self.lib.bo_nn_b1_forward(self._b1, x.data_ptr(), y.data_ptr(), B, torch.cuda.current_stream(x.device).cuda_stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return y

    def check_b1(self):
        """Raise if a hand-off wait inside the last tower_b1 launch gave up (its bounded spin ran out: the evaluation is invalid).
        Synchronises torch's current stream."""
        t = self.__dict__.get("_b1")
        if not t:
            return
        code = C.c_int32(0)
        if self.lib.bo_nn_b1_status(t, C.byref(code), torch.cuda.current_stream(self._b1_dev).cuda_stream):
            raise E.EngineError(self.lib.bo_last_error().decode())
        if code.value < 0:
            raise E.EngineError("tower_b1: an activation left the fp16 range (|v| > 65504) and was saturated -- the evaluation is wrong on the fp16 "
                                "matrix pipe; run this net with BETAONE_F32_TOWER=fp32 (FusedPolicyValueNet(..., f32_pipe=True))")
        if code.value:
            raise E.EngineError(f"tower_b1: a hand-off inside the launch timed out (phase {code.value - 1}); the evaluation is invalid")

    def _build_tower_split(self, dev):
        """float32 tower on the fp16 matrix pipe (bo_nn_tower_create, BO_TOWER_SPLIT_F16): (hi, lo) fp16 pairs of the scaled weights,
        the inverse scale behind every bias."""
        c = self.w_in.shape[0]
        if c not in (128, 256) or self.w_in.shape[1] != 120:
            raise E.EngineError("conv='tower_split' supports 120 input planes and 128 or 256 filters")
        # 128 filters: the products as 16x16x32 tiles (csrc/bo_tower_s16.h) or 32x32x16 (bo_tower_s.h); BETAONE_SPLIT_TILE=16 / 32 for A/B runs
        self.split_tile = 16 if (c == 128 and os.environ.get("BETAONE_SPLIT_TILE", SPLIT_TILE_DEFAULT) == "16") else 32
        pack = pack_conv_weight_split16 if self.split_tile == 16 else pack_conv_weight_split
        wts, params, layers = [], [], []
        n_h = n_p = 0  # halves in wts, floats in params

        def add_w(t16):
            nonlocal n_h
            off = n_h // 8
            flat = t16.reshape(-1).numpy()
            wts.append(flat)
            n_h += flat.size
            return off

        def add_p(t):
            nonlocal n_p
            off = n_p
            flat = t.detach().float().cpu().contiguous().reshape(-1).numpy()
            params.append(flat)
            n_p += flat.size
            return off

        def add_conv(w, bias):  # -> (weights offset, bias offset); params: bias [c], 1 / scale, padding to a multiple of 4 floats
            w = w.detach().float().cpu()
            s = split_scale(w)
            return add_w(pack(w, s)), add_p(torch.cat([bias.detach().float().cpu().reshape(-1), torch.tensor([1.0 / s, 0.0, 0.0, 0.0])]))

        w0 = torch.zeros((c, 128, 3, 3))
        w0[:, :120] = self.w_in.detach().float().cpu()
        wo, bo = add_conv(w0, self.b_in)
        layers.append([wo, 9 * 128 // 16, bo, 0, 0, 0, 0, 0])
        for w1, b1, w2, b2, se in self.blocks:
            wo, bo = add_conv(w1, b1)
            layers.append([wo, 9 * c // 16, bo, 1, 0, 0, 0, 0])
            wo, bo = add_conv(w2, b2)
            if se is not None:
                if se[0].shape[0] > 16 or se[0].shape[0] > c // 16:
                    raise E.EngineError("conv='tower_split' supports SE hidden widths up to min(16, filters/16)")
                layers.append([wo, 9 * c // 16, bo, 3, add_p(se[0]), add_p(se[1]), se[0].shape[0], 0])
                if n_p % 4:
                    add_p(torch.zeros(4 - n_p % 4))
            else:
                layers.append([wo, 9 * c // 16, bo, 2, 0, 0, 0, 0])
        layers[-1][7] = 1
        self._head_ch, self._head_split = self.w_head.shape[0], self.n_policy_ch
        mt = (self._head_ch + 31) // 32
        wh = torch.zeros((mt * 32, c))
        wh[:self._head_ch] = self.w_head.detach().float().cpu().reshape(self._head_ch, c)
        hs = split_scale(wh)
        hi, lo = split_f16(wh.double() * hs)
        whp = torch.stack([hi, lo], 0).reshape(2, mt, 32, c // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()  # [mt][st][hl][kg][o][i]
        head = np.array([self._head_ch, self._head_split, add_w(whp),
                         add_p(torch.cat([self.b_head.detach().float().cpu().reshape(-1), torch.tensor([1.0 / hs])]))], dtype=np.int32)
        wts = np.ascontiguousarray(np.concatenate(wts), dtype=np.float16)
        if wts.size % 8:
            wts = np.concatenate([wts, np.zeros(8 - wts.size % 8, np.float16)])
        params = np.ascontiguousarray(np.concatenate(params), dtype=np.float32)
        table = np.ascontiguousarray(np.array(layers, dtype=np.int32))
        handle = C.c_void_p()
        rc = self.lib.bo_nn_tower_create(table.ctypes.data, len(layers), wts.ctypes.data, wts.size // 2, params.ctypes.data, params.size, c,
                                         4 if self.split_tile == 16 else 3, head.ctypes.data, dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(handle))
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        self.c, self._tower, self._tower_dev = c, handle, dev

    def _build_tower_f16(self, dev):
        """fp16 tower (bo_nn_tower_create, BO_TOWER_DIRECT_F16) + half copies of the three head Linear layers."""
        self.f16_tile = 16 if os.environ.get("BETAONE_F16_TILE", F16_TILE_DEFAULT.get(self.w_in.shape[0], "32")) == "16" else 32
        pack16 = pack_conv_weight_f16_t16 if self.f16_tile == 16 else pack_conv_weight_f16
        c = self.w_in.shape[0]
        if c not in (128, 256) or self.w_in.shape[1] != 120:
            raise E.EngineError("conv='tower_f16' supports 120 input planes and 128 or 256 filters")
        wts, params, layers = [], [], []
        n_h = n_p = 0  # halves in wts, floats in params

        def add_w(t16):
            nonlocal n_h
            off = n_h // 8
            flat = t16.reshape(-1).numpy()
            wts.append(flat)
            n_h += flat.size
            return off

        def add_p(t):
            nonlocal n_p
            off = n_p
            flat = t.detach().float().cpu().contiguous().reshape(-1).numpy()
            params.append(flat)
            n_p += flat.size
            return off

        w0 = torch.zeros((c, 128, 3, 3))
        w0[:, :120] = self.w_in.detach().float().cpu()
        layers.append([add_w(pack16(w0)), 9 * 128 // 16, add_p(self.b_in), 0, 0, 0, 0, 0])
        for w1, b1, w2, b2, se in self.blocks:
            layers.append([add_w(pack16(w1.detach().float().cpu())), 9 * c // 16, add_p(b1), 1, 0, 0, 0, 0])
            p2 = add_w(pack16(w2.detach().float().cpu()))
            if se is not None:
                if se[0].shape[0] > 16:
                    raise E.EngineError("conv='tower_f16' supports SE hidden widths up to 16")
                layers.append([p2, 9 * c // 16, add_p(b2), 3, add_p(se[0]), add_p(se[1]), se[0].shape[0], 0])
            else:
                layers.append([p2, 9 * c // 16, add_p(b2), 2, 0, 0, 0, 0])
        layers[-1][7] = 1
        self._head_ch, self._head_split = self.w_head.shape[0], self.n_policy_ch
        mt = (self._head_ch + 31) // 32
        wh = torch.zeros((mt * 32, c))
        wh[:self._head_ch] = self.w_head.detach().float().cpu().reshape(self._head_ch, c)
        whp = wh.reshape(mt, 32, c // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().half()  # [mt][st][kg][o][i]
        head = np.array([self._head_ch, self._head_split, add_w(whp), add_p(self.b_head)], dtype=np.int32)
        wts = np.ascontiguousarray(np.concatenate(wts), dtype=np.float16)
        if wts.size % 2:
            wts = np.concatenate([wts, np.zeros(1, np.float16)])
        params = np.ascontiguousarray(np.concatenate(params), dtype=np.float32)
        table = np.ascontiguousarray(np.array(layers, dtype=np.int32))
        handle = C.c_void_p()
        rc = self.lib.bo_nn_tower_create(table.ctypes.data, len(layers), wts.ctypes.data, wts.size // 2, params.ctypes.data, params.size, c,
                                         (5 if self.f16_tile == 16 else 2), head.ctypes.data, dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(handle))
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        self.c, self._tower, self._tower_dev = c, handle, dev
        import copy

        self.policy_fc_h = copy.deepcopy(self.policy_fc).half()
        self.value_fc1_h = copy.deepcopy(self.value_fc1).half()
        self.value_fc2_h = copy.deepcopy(self.value_fc2).half()
        self.wants_float32_input = True

    def _tower_f16_forward(self, x):
        if x.dtype != torch.float32:
            x = x.float()
        if x.device != self._tower_dev or x.shape[1:] != (120, 8, 8):
            raise E.EngineError("tower_f16: x must be [B, 120, 8, 8] on the tower's device")
        x = x.contiguous()
        B = x.shape[0]
        pa = torch.empty((B, self._head_split * 64), dtype=torch.float16, device=x.device)
        pb = torch.empty((B, (self._head_ch - self._head_split) * 64), dtype=torch.float16, device=x.device)
        rc = self.lib.bo_nn_tower_forward(self._tower, x.data_ptr(), None, pa.data_ptr(), pb.data_ptr(), B,
                                          torch.cuda.current_stream(x.device).cuda_stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return pa, pb

    def _build_tower(self, dev, winograd=False):
        """Flatten the trunk into the three host arrays of bo_nn_tower_create (include/betaone_engine.h)."""
        c = self.w_in.shape[0]
        if c not in (64, 128) or self.w_in.shape[1] != 120:
            raise E.EngineError("conv='tower' supports 120 input planes and 64 or 128 filters")
        wts, params, layers = [], [], []
        n_w = n_p = 0

        def add_w(w):
            nonlocal n_w
            off = n_w // 4
            flat = (pack_conv_weight_winograd if winograd else pack_conv_weight)(w.detach().float().cpu()).reshape(-1).numpy()
            wts.append(flat)
            n_w += flat.size
            return off

        def add_p(t):
            nonlocal n_p
            off = n_p
            flat = t.detach().float().cpu().contiguous().reshape(-1).numpy()
            params.append(flat)
            n_p += flat.size
            return off

        w0 = torch.zeros((c, 128, 3, 3))
        w0[:, :120] = self.w_in.detach().float().cpu()
        ksteps = (lambda cin: cin // 4) if winograd else (lambda cin: cin // 8)
        layers.append([add_w(w0), ksteps(128), add_p(self.b_in), 0, 0, 0, 0, 0])
        for w1, b1, w2, b2, se in self.blocks:
            layers.append([add_w(w1), ksteps(c), add_p(b1), 1, 0, 0, 0, 0])
            if se is not None:
                if se[0].shape[0] > 16:
                    raise E.EngineError("conv='tower' supports SE hidden widths up to 16")
                layers.append([add_w(w2), ksteps(c), add_p(b2), 3, add_p(se[0]), add_p(se[1]), se[0].shape[0], 0])
            else:
                layers.append([add_w(w2), ksteps(c), add_p(b2), 2, 0, 0, 0, 0])
        layers[-1][7] = 1
        head = None
        if winograd:  # the two 1x1 head convolutions run behind the tower on the LDS-resident output
            self._head_ch, self._head_split = self.w_head.shape[0], self.n_policy_ch
            mb = (self._head_ch + 15) // 16
            wh = torch.zeros((mb * 16, c))
            wh[:self._head_ch] = self.w_head.detach().float().cpu().reshape(self._head_ch, c)
            whp = wh.reshape(mb, 16, c // 16, 4, 4).permute(0, 2, 4, 1, 3).contiguous()  # [mb][g][k][o16][e], ic = 16g + 4e + k
            if n_p % 4:
                add_p(torch.zeros(4 - n_p % 4))
            head = np.array([self._head_ch, self._head_split, add_p(whp), add_p(self.b_head)], dtype=np.int32)
        wts = np.ascontiguousarray(np.concatenate(wts), dtype=np.float32)
        params = np.ascontiguousarray(np.concatenate(params), dtype=np.float32)
        table = np.ascontiguousarray(np.array(layers, dtype=np.int32))
        handle = C.c_void_p()
        rc = self.lib.bo_nn_tower_create(table.ctypes.data, len(layers), wts.ctypes.data, wts.size, params.ctypes.data, params.size, c,
                                         1 if winograd else 0, head.ctypes.data if head is not None else None, dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(handle))
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        self.c, self._tower, self._tower_dev = c, handle, dev

    def __del__(self):
        try:
            t = self.__dict__.get("_tower")
            if t:
                self.__dict__["_tower"] = None
                self.lib.bo_nn_tower_destroy(t)
            t = self.__dict__.get("_b1")
            if t:
                self.__dict__["_b1"] = None
                self.lib.bo_nn_b1_destroy(t)
        except Exception:  # interpreter shutdown
            pass

    def check_overflow(self):
        """conv='tower_split' carries activations as fp16 pairs: raise if any forward since the last check had to saturate one (its
        outputs were wrong) -- bo_nn_tower_status: the word is read and cleared by one atomic exchange on torch's CURRENT stream,
        which must be the stream the forwards were launched on (it is ordered behind them there).  Synchronises that stream.  The
        self-play loop does not call this per ply: it watches the word through the engine's result block (overflow_word_ptr)."""
        if self.conv == "tower_b1":  # (its own two status words: a hand-off that gave up, a saturated activation)
            return self.check_b1()
        t = self.__dict__.get("_tower")
        if not t or self.conv != "tower_split":
            return
        flag = C.c_int32(0)
        rc = self.lib.bo_nn_tower_status(t, C.byref(flag), torch.cuda.current_stream(self._tower_dev).cuda_stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        if flag.value:
            raise E.EngineError(self.OVERFLOW_MESSAGE)

    OVERFLOW_MESSAGE = ("tower_split: an activation left the fp16 range (|v| > 65504) and was saturated -- the evaluations of this net are "
                        "wrong on the fp16 matrix pipe; run it with BETAONE_F32_TOWER=fp32 (best_inference_copy(..., f32_pipe=True))")

    def overflow_words(self):
        """(device address, number of words) of the evaluate stage's own fault words, (0, 1) if it has none: the split-precision tower's
        saturation word, or the one-launch tower's [hand-off timeout code | saturation flag] (conv='tower_b1': what uci.py's searches
        and small self-play batches run on).  Rollout and dropin.mcts hand them to bo_engine_watch_words, so every fetched result
        block brings them along and an invalid evaluation stops the run where it happened."""
        if self.conv == "tower_b1" and self.__dict__.get("_b1"):
            p = C.c_void_p()
            if self.lib.bo_nn_b1_word(self._b1, C.byref(p)):
                raise E.EngineError(self.lib.bo_last_error().decode())
            return (p.value or 0), 2
        return self.overflow_word_ptr(), 1

    def overflow_word_ptr(self) -> int:
        """Device address of the split-precision tower's status word (0 for every other evaluate stage)."""
        t = self.__dict__.get("_tower")
        if not t or self.conv != "tower_split":
            return 0
        p = C.c_void_p()
        if self.lib.bo_nn_tower_word(t, C.byref(p)):
            raise E.EngineError(self.lib.bo_last_error().decode())
        return p.value or 0

    def _tower_forward(self, x, heads=False):
        """Tower output [B, C, 8, 8]; with heads=True (conv='tower_wg') the ReLU'd policy / value planes, flattened."""
        if x.device != self._tower_dev or x.dtype != torch.float32 or x.shape[1:] != (120, 8, 8):
            raise E.EngineError("tower: x must be float32 [B, 120, 8, 8] on the tower's device")
        x = x.contiguous()
        B = x.shape[0]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if heads:
            pa = torch.empty((B, self._head_split * 64), dtype=torch.float32, device=x.device)
            pb = torch.empty((B, (self._head_ch - self._head_split) * 64), dtype=torch.float32, device=x.device)
            timed = self.__dict__.get("tower_events")  # measurement hook (bench.py): a list -> one (start, end) event pair per launch
            if timed is not None:
                timed.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                timed[-1][0].record()
            tbuf = self.__dict__.get("tower_timing_buf")  # (Rollout(time_tower=True): the kernel notes its own duration, also inside graphs)
            if tbuf is not None and self.conv == "tower_split":
                rc = self.lib.bo_nn_tower_forward_timed(self._tower, x.data_ptr(), None, pa.data_ptr(), pb.data_ptr(), B, tbuf.data_ptr(), stream)
            else:
                rc = self.lib.bo_nn_tower_forward(self._tower, x.data_ptr(), None, pa.data_ptr(), pb.data_ptr(), B, stream)
            if timed is not None:
                timed[-1][1].record()
            out = (pa, pb)
        else:
            y = torch.empty((B, self.c, 8, 8), dtype=torch.float32, device=x.device)
            wg = self.conv in ("tower_wg", "tower_split")
            dummy = torch.empty((2, B, self._head_ch * 64), dtype=torch.float32, device=x.device) if wg else None
            rc = self.lib.bo_nn_tower_forward(self._tower, x.data_ptr(), y.data_ptr(), dummy[0].data_ptr() if wg else None,
                                              dummy[1].data_ptr() if wg else None, B, stream)
            out = y
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return out

    def _heads(self, p, v, probs, tail=False):
        """tail=True: stop behind the first launch -> (logits, value_fc1's partial sums [16, B, 256]); Engine.step_heads finishes the
        row (softmax, value) in the step kernel that consumes it (bo_nn_heads flags 4)."""
        B = p.shape[0]
        dev = p.device
        # value_fc1 partial sums: allocated per call -- under graph capture it then comes from the graph's own pool (a module-wide
        # scratch that grew with a later, larger batch left earlier captured graphs writing into freed memory, and was shared
        # by concurrent streams); the caching allocator makes the eager cost a free-list lookup
        scr = torch.empty(4096 * B, dtype=torch.float32, device=dev)
        out = torch.empty((B, 4672), dtype=torch.float32, device=dev)
        value = None if tail else torch.empty((B, 1), dtype=torch.float32, device=dev)
        rc = self.lib.bo_nn_heads(p.data_ptr(), v.data_ptr(), self.policy_fc.weight.data_ptr(), self.policy_fc.bias.data_ptr(),
                                  self.value_fc1.weight.data_ptr(), self.value_fc1.bias.data_ptr(), self.value_fc2.weight.data_ptr(),
                                  self.value_fc2.bias.data_ptr(), out.data_ptr(), None if tail else value.data_ptr(), scr.data_ptr(), B,
                                  (4 if tail else 1 if probs else 0) | (2 if p.dtype == torch.float16 else 0),
                                  torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return (out, scr.view(16, B, 256)) if tail else (out, value)

    def tail_supported(self, batch: int) -> bool:
        """forward_tail exists for this net at this batch: the evaluate stage ends in bo_nn_heads (the reference's head shapes,
        float32 head weights; behind the fp16 tower up to 1024 boards)."""
        ok = self.fused_heads and self.policy_fc.weight.shape == (4672, 128) and self.value_fc1.weight.shape == (256, 2048)
        return bool(ok and (self.conv != "tower_f16" or batch <= 1024))

    def tail_params(self):
        """(value_fc1.bias, value_fc2.weight, value_fc2.bias) device pointers for Engine.step_heads (a captured graph holds copies of
        them as it holds every other weight address of this module: Rollout.swap_model drops its graphs and reads these again)."""
        return self.value_fc1.bias.data_ptr(), self.value_fc2.weight.data_ptr(), self.value_fc2.bias.data_ptr()

    @torch.no_grad()
    def forward_tail(self, x):
        """(logits [B, 4672], partial sums of value_fc1 [16, B, 256]): the evaluate stage without its last launch."""
        if not self.tail_supported(x.shape[0]):
            raise E.EngineError(f"forward_tail: not available for conv={self.conv!r} at batch {x.shape[0]}")
        return self.forward(x, tail=True)

    def _heads_f16(self, p, v, probs):
        B, dev = p.shape[0], p.device
        out = torch.empty((B, 4672), dtype=torch.float32, device=dev)
        value = torch.empty((B, 1), dtype=torch.float32, device=dev)
        scr = torch.empty(20 * B, dtype=torch.float32, device=dev)  # (max, sum of exp) per board and output range (per call: see _heads)
        rc = self.lib.bo_nn_heads_f16(p.data_ptr(), v.data_ptr(), self.policy_fc_h.weight.data_ptr(), self.policy_fc.bias.data_ptr(),
                                      self.value_fc1_h.weight.data_ptr(), self.value_fc1.bias.data_ptr(), self.value_fc2.weight.data_ptr(),
                                      self.value_fc2.bias.data_ptr(), out.data_ptr(), value.data_ptr(), scr.data_ptr(), B, 1 if probs else 0,
                                      torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return out, value

    def _value_tail(self, h):
        out = torch.empty((h.shape[0], 1), dtype=torch.float32, device=h.device)
        rc = self.lib.bo_nn_value_tail(h.data_ptr(), self.value_fc2.weight.data_ptr(), self.value_fc2.bias.data_ptr(), out.data_ptr(), h.shape[0],
                                       h.shape[1], torch.cuda.current_stream(h.device).cuda_stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return out

    def _epi(self, x, bias, res=None):
        B, C = x.shape[0], x.shape[1]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = self.lib.bo_nn_bias_act(x.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, B, C, stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return x

    def _se(self, x, bias, se, res):
        B, C = x.shape[0], x.shape[1]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if self.conv in ("mfma_small", "tower_b1") and C % 16 == 0 and se[0].shape[0] <= 16:  # few boards: C/16 workgroups per board, result into `res`
            rc = self.lib.bo_nn_se_residual_small(x.data_ptr(), bias.data_ptr(), se[0].data_ptr(), se[1].data_ptr(), res.data_ptr(), B, C,
                                                  se[0].shape[0], stream)
            if rc:
                raise E.EngineError(self.lib.bo_last_error().decode())
            return res
        rc = self.lib.bo_nn_se_residual(x.data_ptr(), bias.data_ptr(), se[0].data_ptr(), se[1].data_ptr(), res.data_ptr(), B, C,
                                        se[0].shape[0], stream)
        if rc:
            raise E.EngineError(self.lib.bo_last_error().decode())
        return x

    def _tower_mfma(self, x):
        L, C = self.lib, self.c
        x = conv3x3_mfma(L, x.contiguous(), self.p_in, self.b_in, C, 1)
        for (w1, b1, w2, b2, se), (p1, p2) in zip(self.blocks, self.packed):
            y = conv3x3_mfma(L, x, p1, b1, C, 1)
            if se is not None:
                x = self._se(conv3x3_mfma(L, y, p2, self.zero_bias, C, 0), b2, se, x)
            else:
                x = conv3x3_mfma(L, y, p2, b2, C, 2, residual=x)
        return x

    def _tower_small(self, x):
        L, C = self.lib, self.c
        x = conv3x3_small(L, x.contiguous(), self.p_in, self.b_in, 128, C, 1)
        for (w1, b1, w2, b2, se), (p1, p2) in zip(self.blocks, self.packed):
            y = conv3x3_small(L, x, p1, b1, C, C, 1)
            if se is not None:
                x = self._se(conv3x3_small(L, y, p2, self.zero_bias, C, C, 0), b2, se, x)
            else:
                x = conv3x3_small(L, y, p2, b2, C, C, 2, residual=x)
        return x

    @torch.no_grad()
    def forward_probs(self, x):
        """(softmax(logits, dim=1) in float32, value): the policy softmax of mcts.py:185,287 issued where it costs least."""
        return self.forward(x, probs=True)

    @torch.no_grad()
    def forward(self, x, probs: bool = False, tail: bool = False):
        if self.conv == "tower_f16":
            p, v = self._tower_f16_forward(x)
            if self.fused_heads and p.shape[1] == 128 and v.shape[1] == 2048:
                # Up to 1024 boards (configs[4]: 512): fp16 head planes widened on load; float32 head weights, accumulation, softmax and
                # value on the fp32 matrix pipe (closer to the float32 net than the half-precision GEMMs; tile-parallel, so a few
                # hundred rows still fill the chip).  Beyond (fast mode: 4 096 .. 131 072 rows): fp16 weights on the fp16 pipe, the
                # policy FC and its softmax fused per 32-board wave, the value head in one kernel (bo_heads.h) -- no library launch.
                if p.shape[0] <= 1024:
                    return self._heads(p, v, probs, tail)
                if os.environ.get("BETAONE_HEADS_F16", "1") != "0":  # (0: the library path below, for A/B runs)
                    return self._heads_f16(p, v, probs)
            logits = self.policy_fc_h(p)
            return (torch.softmax(logits.float(), dim=1) if probs else logits), torch.tanh(self.value_fc2_h(F.relu(self.value_fc1_h(v))))
        if self.conv in ("tower_wg", "tower_split"):  # tower + head convolutions in one kernel, then the heads
            p, v = self._tower_forward(x, heads=True)
            if self.fused_heads and p.shape[1] == 128 and v.shape[1] == 2048:
                return self._heads(p, v, probs, tail)  # policy FC + softmax + value head: two launches (csrc/bo_heads.h)
            # the value head (2 small kernels) runs beside the policy GEMM: a fork/join of streams, also inside a captured graph
            cur = torch.cuda.current_stream(x.device)
            side = self.__dict__.get("_side")
            if side is None or side.device != x.device:
                side = self.__dict__["_side"] = torch.cuda.Stream(x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                h = torch._addmm_activation(self.value_fc1.bias, v, self.value_fc1.weight.t())  # relu(fc1)
                value = self._value_tail(h)
            logits = self.policy_fc(p)
            if probs:  # before the join: the softmax runs beside the value head instead of behind the streams' rendezvous
                logits = torch.softmax(logits.float(), dim=1)
            cur.wait_stream(side)
            return logits, value
        if self.conv in ("mfma", "tower", "mfma_small", "tower_b1"):
            x = (self._tower_mfma(x) if self.conv == "mfma" else self._tower_small(x) if self.conv == "mfma_small"
                 else self._tower_b1(x) if self.conv == "tower_b1" else self._tower_forward(x))
            h = self._epi(F.conv2d(x, self.w_head, None), self.b_head)
            p = h[:, :self.n_policy_ch].flatten(1)
            v = h[:, self.n_policy_ch:].flatten(1)
            if self.fused_heads and h.dtype == torch.float32 and p.shape[1] == 128 and v.shape[1] == 2048:
                # (batch 1 -- uci.py's analysis -- : the slices are contiguous already; this replaces nine small launches)
                return self._heads(p.contiguous(), v.contiguous(), probs, tail)
            logits = self.policy_fc(p)
            return (torch.softmax(logits.float(), dim=1) if probs else logits), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
        x = self._epi(F.conv2d(x, self.w_in, None, padding=1), self.b_in)
        for w1, b1, w2, b2, se in self.blocks:
            y = self._epi(F.conv2d(x, w1, None, padding=1), b1)
            y = F.conv2d(y, w2, None, padding=1)
            x = self._se(y, b2, se, x) if se is not None else self._epi(y, b2, x)
        h = self._epi(F.conv2d(x, self.w_head, None), self.b_head)
        p = h[:, :self.n_policy_ch].flatten(1)
        v = h[:, self.n_policy_ch:].flatten(1)
        logits = self.policy_fc(p)
        return (torch.softmax(logits.float(), dim=1) if probs else logits), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
