// betaone_amd/csrc/bo_tower.h -- the whole residual tower of the evaluate stage as ONE persistent kernel (gfx950).
//
// PolicyValueNet's trunk (/root/reference/network.py:48-118,167-190: input conv, N residual blocks, some with an SE
// gate) applied to 8x8 boards is 2N+1 3x3 convolutions whose activations (C x 64 floats per board = 32 KB at C=128)
// fit in LDS.  A workgroup therefore keeps ONE board on its CU for the whole tower:
//
//   * LDS holds two zero-padded 10x10 images in the channel-interleaved layout of bo_conv.h: P (the staged input
//     planes, later the mid activation of a block) and Q (the block input / skip connection).  Layer epilogues write
//     straight into the other image, so activations never leave the CU between the first load of the input planes
//     and the final store of the tower output; the halos are zeroed once per kernel;
//   * each convolution is the implicit GEMM of bo_conv.h (v_mfma_f32_32x32x2_f32, wave w = output channels
//     32w..32w+31 x all 64 squares, A fragments = one global_load_dwordx4 per (tap, 8 channels) from the L2-resident
//     packed weights, B operands = one ds_read_b128); weight fragments are prefetched one channel group ahead across
//     layer (and board) boundaries, so the matrix cores only idle for the epilogue + one barrier per layer;
//   * bias, ReLU, skip add and the SE gate (channel means -> FC -> ReLU -> FC -> sigmoid, network.py:33-45) are
//     computed on the accumulator registers;
//   * with 256 boards per batch (BASELINE.json's 256 concurrent games) every CU of the MI355X owns exactly one board.
//     Larger batches loop (grid = min(B, #CU)), smaller ones leave CUs idle (nn_tune.py then prefers MIOpen).
// MFMA time at 100 % issue, C=128, 8+2 blocks: 21 layers x 73.7k cycles = 0.645 ms per 256 boards.
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_conv.h"

struct bo_tower_layer {
    int w_off4;     // offset of the packed weights [9][t4][C][2] float4, in float4 units
    int t4;         // input channel groups of 8 (even; the input conv is zero-padded from 120 to 128 channels)
    int bias_off;   // offset of bias[C] in params
    int kind;       // 0 input conv (P -> Q), 1 first conv of a block (Q -> P), 2 second conv + skip (P,Q -> Q), 3 = 2 with SE gate
    int se_w1_off;  // kind 3: W1 [H][C] in params
    int se_w2_off;  // kind 3: W2 [C][H] in params
    int hidden;     // kind 3: H (<= 16)
    int last;       // 1: also store the result to y (NCHW)
};

// Cross-lane sums on the VALU (DPP), not through LDS (ds_bpermute): after bo_row_sum every lane holds the sum of its
// 16-lane row; bo_half_sum: lanes 16..31 / 48..63 hold the sum of lanes 0..31 / 32..63; bo_wave_sum63: lane 63 holds all 64.
#define BO_DPP(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false))
__device__ inline float bo_row_sum(float v) {
    v += BO_DPP(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
    v += BO_DPP(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
    v += BO_DPP(v, 0x141, 0xF);  // row_half_mirror
    v += BO_DPP(v, 0x140, 0xF);  // row_mirror
    return v;
}
__device__ inline float bo_half_sum(float v) {
    v = bo_row_sum(v);
    v += BO_DPP(v, 0x142, 0xA);  // row_bcast15 into rows 1 and 3
    return v;
}
__device__ inline float bo_wave_sum63(float v) {
    v = bo_half_sum(v);
    v += BO_DPP(v, 0x143, 0xC);  // row_bcast31 into rows 2 and 3
    return v;
}

// the two 1x1 head convolutions fused behind the tower (bo_tower_wg.h): channels [0, split) go to out_a [B][split][64],
// channels [split, channels) to out_b [B][channels - split][64]; weights [channels][C] and bias in params
struct bo_tower_head {
    int channels = 0, split = 0, w_off = 0, b_off = 0;
    float *out_a = nullptr, *out_b = nullptr;
};

template <int C>
__global__ void __launch_bounds__(C * 2)
bo_k_tower(const float *__restrict__ x, const bo_f32x4 *__restrict__ wts, const float *__restrict__ params,
           const bo_tower_layer *__restrict__ layers, int n_layers, float *__restrict__ y, int B) {
    constexpr int PITCH = 100, NT = C * 2, GP = 16, GQ = C / 8, CIN0 = 120;
    __shared__ bo_f32x4 P[GP * 2 * PITCH];
    __shared__ bo_f32x4 Q[GQ * 2 * PITCH];
    __shared__ float pooled[C];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), k = lane >> 5, j = lane & 31;
    const int cell = ((j >> 3) + 1) * 10 + (j & 7) + 1;  // padded cell of square j; square j+32 is cell+40
    const int oc_a = wave * 32 + j;                      // A operand row of this lane

    for (int i = tid; i < GP * 2 * PITCH; i += NT) P[i] = bo_f32x4{0, 0, 0, 0};
    for (int i = tid; i < GQ * 2 * PITCH; i += NT) Q[i] = bo_f32x4{0, 0, 0, 0};

    bo_f32x4 fa[9], fb[9], bn0, bn1;
    bo_f32x16 acc0, acc1;
    const unsigned wlane = (unsigned)oc_a * 2 + k;  // lane part of a weight address; the rest is uniform
    auto load = [&](bo_f32x4(&a)[9], int w_off4, int nt4, int t4) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++) a[tap] = (wts + w_off4 + (size_t)(tap * nt4 + t4) * C * 2)[wlane];
    };
    auto read_b = [&](const bo_f32x4 *xl, int t4, int tap) {
        const int off = (tap / 3 - 1) * 10 + (tap % 3 - 1);
        bn0 = xl[t4 * 2 * PITCH + off];
        bn1 = xl[t4 * 2 * PITCH + off + 40];
    };
    // 72 MFMAs of channel group t4 with fragments a.  In the shadow of each tap's 8 MFMAs: the LDS reads of the next
    // tap and the load of the same tap of the next group (layer w_off4n / nt4n, group t4n) into an (see bo_conv.h).
    auto compute = [&](const bo_f32x4(&a)[9], bo_f32x4(&an)[9], const bo_f32x4 *xl, int t4, int t4_next_b, int w_off4n, int nt4n, int t4n) {
        const bo_f32x4 *wn = wts + w_off4n + (size_t)t4n * C * 2;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const bo_f32x4 b0 = bn0, b1 = bn1;
            if (tap < 8) read_b(xl, t4, tap + 1);
            else read_b(xl, t4_next_b, 0);
            an[tap] = (wn + (size_t)(tap * nt4n) * C * 2)[wlane];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap][e], b0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap][e], b1[e], acc1, 0, 0, 0);
            }
            BO_CONV_TAP_SCHEDULE();
        }
    };
    // float2 slot of output channels (q, s): oc = 32*wave + 8q + 4k + s (.x) and oc + 2 (.y), at this lane's square
    auto slot = [&](bo_f32x4 *buf, int q, int s, int half) -> float2 * {
        return reinterpret_cast<float2 *>(reinterpret_cast<float *>(buf) +
                                          ((((wave * 4 + q) * 2 + s) * PITCH + cell + 40 * half) * 4 + 2 * k));
    };

    load(fa, layers[0].w_off4, layers[0].t4, 0);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- stage the 120 input planes of board b into P (channels 120..127 of the padded input conv are zero) ----
        {
            const bo_f32x4 *xb = reinterpret_cast<const bo_f32x4 *>(x + (size_t)b * CIN0 * 64);
            float *Pf = reinterpret_cast<float *>(P);
            __syncthreads();  // the zero fill / the previous board's readers of P are done
            for (int i = tid; i < 128 * 16; i += NT) {
                const bo_f32x4 v = i < CIN0 * 16 ? xb[i] : bo_f32x4{0, 0, 0, 0};
                const int ic = i >> 4, q = i & 15, r = q >> 1, c0 = (q & 1) * 4;
                float *dst = Pf + (((ic >> 3) * 2 + (ic & 1)) * PITCH + (r + 1) * 10 + c0 + 1) * 4 + ((ic & 7) >> 1);
                dst[0] = v[0]; dst[4] = v[1]; dst[8] = v[2]; dst[12] = v[3];
            }
            __syncthreads();
        }
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = layers[l];
            const bo_tower_layer Ln = layers[l + 1 < n_layers ? l + 1 : 0];
            const bo_f32x4 *xl = (L.kind == 1 ? Q : P) + k * PITCH + cell;
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) bv[r] = params[L.bias_off + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * k];
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
            read_b(xl, 0, 0);
            for (int t4 = 0; t4 < L.t4; t4 += 2) {
                compute(fa, fb, xl, t4, t4 + 1, L.w_off4, L.t4, t4 + 1);
                const bool more = t4 + 2 < L.t4;  // otherwise prefetch the first group of the next layer
                compute(fb, fa, xl, t4 + 1, more ? t4 + 2 : t4 + 1, more ? L.w_off4 : Ln.w_off4, more ? L.t4 : Ln.t4, more ? t4 + 2 : 0);
            }

            // ---- epilogue on the accumulators: D row = (r&3) + 8*(r>>2) + 4k, col = j ----
            if (L.kind <= 1) {
                bo_f32x4 *out = L.kind == 0 ? Q : P;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int r = 4 * q + s;
                        float2 v0 = {acc0[r] + bv[r], acc0[r + 2] + bv[r + 2]}, v1 = {acc1[r] + bv[r], acc1[r + 2] + bv[r + 2]};
                        v0.x = fmaxf(v0.x, 0.0f); v0.y = fmaxf(v0.y, 0.0f); v1.x = fmaxf(v1.x, 0.0f); v1.y = fmaxf(v1.y, 0.0f);
                        *slot(out, q, s, 0) = v0;
                        *slot(out, q, s, 1) = v1;
                    }
            } else {
                float gate[16];
#pragma unroll
                for (int r = 0; r < 16; r++) gate[r] = 1.0f;
                if (L.kind == 3) {
                    // channel means of conv + bias over the 64 squares (AdaptiveAvgPool2d(1))
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float s = acc0[r] + acc1[r];
#pragma unroll
                        for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m);
                        if (j == 0) pooled[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * k] = s * (1.0f / 64.0f) + bv[r];
                    }
                    __syncthreads();
                    // hidden = relu(W1 mean) (every wave, redundantly), gate = sigmoid(W2 hidden) for channel 32*wave + j
                    const float *w1 = params + L.se_w1_off, *w2 = params + L.se_w2_off;
                    float a = 0.0f;
#pragma unroll
                    for (int h = 0; h < 16; h++) {
                        if (h < L.hidden) {
                            float p = 0.0f;
                            for (int c = lane; c < C; c += 64) p += w1[h * C + c] * pooled[c];
#pragma unroll
                            for (int m = 32; m >= 1; m >>= 1) p += __shfl_xor(p, m);
                            a += w2[oc_a * L.hidden + h] * fmaxf(p, 0.0f);
                        }
                    }
                    const float g = 1.0f / (1.0f + expf(-a));
#pragma unroll
                    for (int r = 0; r < 16; r++) gate[r] = __shfl(g, (r & 3) + 8 * (r >> 2) + 4 * k);
                }
                float *yb = y + (size_t)b * C * 64;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int r = 4 * q + s;
                        float2 *s0 = slot(Q, q, s, 0), *s1 = slot(Q, q, s, 1);
                        const float2 r0 = *s0, r1 = *s1;
                        float2 v0, v1;
                        if (L.kind == 3) {
                            v0 = {(acc0[r] + bv[r]) * gate[r] + r0.x, (acc0[r + 2] + bv[r + 2]) * gate[r + 2] + r0.y};
                            v1 = {(acc1[r] + bv[r]) * gate[r] + r1.x, (acc1[r + 2] + bv[r + 2]) * gate[r + 2] + r1.y};
                        } else {
                            v0 = {acc0[r] + bv[r] + r0.x, acc0[r + 2] + bv[r + 2] + r0.y};
                            v1 = {acc1[r] + bv[r] + r1.x, acc1[r + 2] + bv[r + 2] + r1.y};
                        }
                        v0.x = fmaxf(v0.x, 0.0f); v0.y = fmaxf(v0.y, 0.0f); v1.x = fmaxf(v1.x, 0.0f); v1.y = fmaxf(v1.y, 0.0f);
                        *s0 = v0;
                        *s1 = v1;
                        if (L.last) {
                            const int o = wave * 32 + 8 * q + 4 * k + s;
                            yb[o * 64 + j] = v0.x; yb[o * 64 + 32 + j] = v1.x;
                            yb[(o + 2) * 64 + j] = v0.y; yb[(o + 2) * 64 + 32 + j] = v1.y;
                        }
                    }
            }
            __syncthreads();
        }
    }
}
#endif
