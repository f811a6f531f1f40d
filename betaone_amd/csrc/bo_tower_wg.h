// betaone_amd/csrc/bo_tower_wg.h -- the residual tower as one persistent kernel, Winograd F(2x2, 3x3) on the fp32
// matrix cores (gfx950).  Same contract as bo_tower.h (one workgroup keeps one board in LDS for the whole tower of
// /root/reference/network.py:48-118,167-190); the 3x3 convolutions cost 16 multiplies per 2x2 output tile instead
// of 36 (2.25x fewer MFMA cycles than the direct form, which runs at 88 % of the fp32 MFMA ceiling).
//
//   Y = A^T [ sum_ic (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// An 8x8 board is 16 tiles, so for each of the 16 Winograd positions the channel sum is a GEMM
// [C_out x C_in] x [C_in x 16 tiles]: v_mfma_f32_16x16x4_f32 with M = 16 output channels, N = the 16 tiles, K = 4
// input channels.  The operand layouts make both transforms register-local:
//   * B operand lane (k = lane>>4, n = lane&15) is input channel 4*step + k of tile n: the lane reads that 4x4 patch
//     from the zero-padded 10x10 LDS image (8 ds_read_b64), applies B^T d B (32 adds) and holds all 16 positions of
//     its (channel, tile) -- one B register per position;
//   * D lane (n = lane&15, rows 4*(lane>>4)+r) holds, over the 16 position accumulators, the complete 4x4 M matrix of
//     tile n for 4 output channels: A^T M A (24 adds), bias, ReLU, skip and SE gate happen in registers and the 2x2
//     result goes straight into the other LDS image;
//   * A operands: the host stores U = G g G^T as Up[step][oc/16][pos/4][lane = 16*(ic&3) + (oc&15)][pos&3], one
//     coalesced global_load_dwordx4 per 4 positions, prefetched one K-step ahead.
// A wave owns 16 output channels (16 position accumulators of 4 registers); C/16 waves per workgroup = two per SIMD at
// C = 128, which keeps every wave under 256 registers (four weight sets in flight without spilling accumulators into
// copies) and lets one wave's MFMAs cover the other's waits.  A K-step is 16 MFMAs of 32 cycles per wave.  MFMA time
// at 100 % issue, C = 128: 32 steps x 2 waves x 16 x 32 cycles = 32.8k cycles = 13.7 us per layer (direct form: 30.7 us).  fp32 throughout; the transforms use only +/-/x0.5, results agree with the direct form to
// ~1e-6 relative.
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_tower.h"

// LAB (scripts/conv_lab.hip only): 0 = the kernel; 1 = no weight loads; 2 = no LDS patch reads
template <int C, int LAB = 0>
__global__ void __launch_bounds__(C * 4)
bo_k_tower_wg(const float *__restrict__ x, const bo_f32x4 *__restrict__ wts, const float *__restrict__ params,
              const bo_tower_layer *__restrict__ layers, int n_layers, float *__restrict__ y, int B) {
    constexpr int IMG = 100, NT = C * 4, CP = 128, CIN0 = 120, OB = C / 16;
    __shared__ __attribute__((aligned(16))) float P[CP * IMG];  // staged input planes / mid activation of a block
    __shared__ __attribute__((aligned(16))) float Q[C * IMG];   // block input = skip connection
    __shared__ float pooled[C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int ty = n >> 2, tx = n & 3;
    const int patch = 20 * ty + 2 * tx;       // top-left of the tile's 4x4 input patch in a padded 10x10 image
    const int ocell = patch + 11;             // top-left of its 2x2 output block

    for (int i = tid; i < CP * IMG; i += NT) P[i] = 0.0f;
    for (int i = tid; i < C * IMG; i += NT) Q[i] = 0.0f;

    bo_f32x4 acc[16];
    bo_f32x4 a0[4], a1[4], a2[4], a3[4];  // A fragments of four consecutive K-steps: [position quad]
    float d[16], v0[16], v1[16];   // raw patch of the next step; transformed patches of two consecutive steps
    // weights of one K-step of a layer: uniform base + (oc block, position quad, lane)
    auto load_w = [&](bo_f32x4(&a)[4], int w_off4, int step) {
        const bo_f32x4 *wb = wts + w_off4 + (size_t)step * (OB * 4 * 64) + (wave * 4) * 64 + lane;
#pragma unroll
        for (int pq = 0; pq < 4; pq++) a[pq] = wb[pq * 64];
    };
    auto load_d = [&](const float *img, int step) {
        const float *p = img + (4 * step + kq) * IMG + patch;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float2 lo = *reinterpret_cast<const float2 *>(p + 10 * i), hi = *reinterpret_cast<const float2 *>(p + 10 * i + 2);
            d[4 * i] = lo.x; d[4 * i + 1] = lo.y; d[4 * i + 2] = hi.x; d[4 * i + 3] = hi.y;
        }
    };
    float w[16];
    auto xform_rows = [&]() {  // w = B^T d
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w[j] = d[j] - d[8 + j];
            w[4 + j] = d[4 + j] + d[8 + j];
            w[8 + j] = d[8 + j] - d[4 + j];
            w[12 + j] = d[4 + j] - d[12 + j];
        }
    };
    auto xform_cols = [&](float(&v)[16]) {  // V = w B
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[4 * i] = w[4 * i] - w[4 * i + 2];
            v[4 * i + 1] = w[4 * i + 1] + w[4 * i + 2];
            v[4 * i + 2] = w[4 * i + 2] - w[4 * i + 1];
            v[4 * i + 3] = w[4 * i + 1] - w[4 * i + 3];
        }
    };
    // One K-step = 16 MFMAs in four blocks of 4 (position quad pq).  Each block also issues the weight load of the same
    // quad two steps ahead; blocks 0-1 read the next step's patch from LDS, blocks 2-3 transform it.  The
    // sched_group_barriers put every one of those instructions into the shadow of a different MFMA (the scheduler's own
    // order issues the loads right before their use and exposes their latency).
#define BO_WG_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define BO_WG_BLOCK_LOADS()                                                                         \
    do {                                                                                            \
        BO_WG_SGB(0x008, 1); BO_WG_SGB(0x020, 1); BO_WG_SGB(0x008, 1); BO_WG_SGB(0x100, 2);         \
        BO_WG_SGB(0x008, 1); BO_WG_SGB(0x100, 2); BO_WG_SGB(0x008, 1);                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#define BO_WG_BLOCK_XFORM()                                                                         \
    do {                                                                                            \
        BO_WG_SGB(0x008, 1); BO_WG_SGB(0x020, 1); BO_WG_SGB(0x008, 1); BO_WG_SGB(0x002, 6);         \
        BO_WG_SGB(0x008, 1); BO_WG_SGB(0x002, 6); BO_WG_SGB(0x008, 1); BO_WG_SGB(0x002, 6);         \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
    auto kstep = [&](const bo_f32x4(&a)[4], const float(&v)[16], bo_f32x4(&an)[4], float(&vn)[16], const float *img, int step_d,
                     int w_off4n, int step_w) {
        const bo_f32x4 *wb = wts + w_off4n + (size_t)step_w * (OB * 4 * 64) + (wave * 4) * 64 + lane;
        const float *p = img + (4 * step_d + kq) * IMG + patch;
#pragma unroll
        for (int pq = 0; pq < 4; pq++) {
            if (LAB != 1) an[pq] = wb[pq * 64];
            if (LAB == 2) {
            } else if (pq < 2) {
#pragma unroll
                for (int i = 2 * pq; i < 2 * pq + 2; i++) {
                    const float2 lo = *reinterpret_cast<const float2 *>(p + 10 * i), hi = *reinterpret_cast<const float2 *>(p + 10 * i + 2);
                    d[4 * i] = lo.x; d[4 * i + 1] = lo.y; d[4 * i + 2] = hi.x; d[4 * i + 3] = hi.y;
                }
            } else if (pq == 2) {
                xform_rows();
            } else {
                xform_cols(vn);
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
                acc[4 * pq + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[pq][e], v[4 * pq + e], acc[4 * pq + e], 0, 0, 0);
            if (pq < 2) BO_WG_BLOCK_LOADS();
            else BO_WG_BLOCK_XFORM();
        }
    };

    load_w(a0, layers[0].w_off4, 0);
    load_w(a1, layers[0].w_off4, 1);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- stage the 120 input planes of board b into P (channels 120..127 of the padded input conv are zero) ----
        {
            const bo_f32x4 *xb = reinterpret_cast<const bo_f32x4 *>(x + (size_t)b * CIN0 * 64);
            __syncthreads();  // the zero fill / the previous board's readers of P are done
            for (int i = tid; i < 128 * 16; i += NT) {
                const bo_f32x4 t = i < CIN0 * 16 ? xb[i] : bo_f32x4{0, 0, 0, 0};
                const int ic = i >> 4, q = i & 15, r = q >> 1, c0 = (q & 1) * 4;
                float *dst = P + ic * IMG + (r + 1) * 10 + c0 + 1;
                dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3];
            }
            __syncthreads();
        }
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = layers[l];
            const bo_tower_layer Ln = layers[l + 1 < n_layers ? l + 1 : 0];
            const float *img = L.kind == 1 ? Q : P;
            const int oc0 = wave * 16 + 4 * kq;  // this lane's 4 output channels: oc0 + r
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) bv[r] = params[L.bias_off + oc0 + r];
#pragma unroll
            for (int pos = 0; pos < 16; pos++) acc[pos] = bo_f32x4{0, 0, 0, 0};
            load_d(img, 0);
            xform_rows();
            xform_cols(v0);
            __builtin_amdgcn_sched_barrier(0);
            // Four K-steps per iteration; step s computes with set s%4 while the weights of step s+2 (possibly the first
            // steps of the next layer) are loaded into set (s+2)%4: 16 KB per wave = 64 KB per CU in flight, which is what
            // it takes to stream the L2-resident weights (1 MB per layer per CU) at the rate the matrix cores consume them.
            for (int s = 0; s < L.t4; s += 4) {
                const int t4 = s + 4, t5 = s + 5, nk = L.t4;
                kstep(a0, v0, a2, v1, img, s + 1, L.w_off4, s + 2);
                kstep(a1, v1, a3, v0, img, s + 2, L.w_off4, s + 3);
                kstep(a2, v0, a0, v1, img, s + 3, t4 < nk ? L.w_off4 : Ln.w_off4, t4 < nk ? t4 : t4 - nk);
                kstep(a3, v1, a1, v0, img, t4 < nk ? t4 : s + 3, t5 < nk ? L.w_off4 : Ln.w_off4, t5 < nk ? t5 : t5 - nk);
            }

            // ---- output transform Y = A^T M A per row r: M[i][j] = acc[4i+j][r] ----
            float o[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float t0[4], t1[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    t0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
                    t1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
                }
                o[r][0] = t0[0] + t0[1] + t0[2] + bv[r];
                o[r][1] = t0[1] - t0[2] - t0[3] + bv[r];
                o[r][2] = t1[0] + t1[1] + t1[2] + bv[r];
                o[r][3] = t1[1] - t1[2] - t1[3] + bv[r];
            }
            if (L.kind <= 1) {
                float *out = L.kind == 0 ? Q : P;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float *c = out + (oc0 + r) * IMG + ocell;
                    c[0] = fmaxf(o[r][0], 0.0f); c[1] = fmaxf(o[r][1], 0.0f);
                    c[10] = fmaxf(o[r][2], 0.0f); c[11] = fmaxf(o[r][3], 0.0f);
                }
            } else {
                float gate[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (L.kind == 3) {
                    // channel means of conv + bias over the 64 squares = 16 tiles x 4 outputs (AdaptiveAvgPool2d(1))
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float s = (o[r][0] + o[r][1]) + (o[r][2] + o[r][3]);
#pragma unroll
                        for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
                        if (n == 0) pooled[oc0 + r] = s * (1.0f / 64.0f);
                    }
                    __syncthreads();
                    // hidden = relu(W1 mean) (every wave, redundantly); gate = sigmoid(W2 hidden) for channel 16*wave + (lane&15)
                    const float *w1 = params + L.se_w1_off, *w2 = params + L.se_w2_off;
                    const int oc_g = wave * 16 + n;
                    float a = 0.0f;
#pragma unroll
                    for (int h = 0; h < 16; h++) {
                        if (h < L.hidden) {
                            float p = 0.0f;
                            for (int c = lane; c < C; c += 64) p += w1[h * C + c] * pooled[c];
#pragma unroll
                            for (int m = 32; m >= 1; m >>= 1) p += __shfl_xor(p, m);
                            a += w2[oc_g * L.hidden + h] * fmaxf(p, 0.0f);
                        }
                    }
                    const float g = 1.0f / (1.0f + expf(-a));
#pragma unroll
                    for (int r = 0; r < 4; r++) gate[r] = __shfl(g, 4 * kq + r);
                }
                float *yb = y + (size_t)b * C * 64;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int oc = oc0 + r;
                    float *c = Q + oc * IMG + ocell;
                    float v[4] = {c[0], c[1], c[10], c[11]};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[e] = L.kind == 3 ? o[r][e] * gate[r] + v[e] : o[r][e] + v[e];
                        v[e] = fmaxf(v[e], 0.0f);
                    }
                    c[0] = v[0]; c[1] = v[1]; c[10] = v[2]; c[11] = v[3];
                    if (L.last) {
                        float *g2 = yb + oc * 64 + 16 * ty + 2 * tx;
                        *reinterpret_cast<float2 *>(g2) = float2{v[0], v[1]};
                        *reinterpret_cast<float2 *>(g2 + 8) = float2{v[2], v[3]};
                    }
                }
            }
            __syncthreads();
        }
    }
}
#endif
