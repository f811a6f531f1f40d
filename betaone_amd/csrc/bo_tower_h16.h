// betaone_amd/csrc/bo_tower_h16.h -- bo_k_tower_h (bo_tower_h.h: the residual tower in fp16, two boards per workgroup: BASELINE.json
// configs[4] and the fast mode's evaluate stage) re-tiled for v_mfma_f32_16x16x32_f16, for the reason bo_tower_s16.h gives: a workgroup
// streams every layer's weights through its CU's port to L2 (1.18 MB per 256-filter layer for two boards: 18 us at the port's ~65 GB/s
// against 15.4 us of matrix time), the port delivers bytes per CLOCK, and with every CU multiplying the chip holds 2.10 GHz under
// 16x16x32 tiles where it holds 1.93 under 32x32x16 (profiles/r05_tower_bound.md section 3).  Same contract, same operand bytes, same
// accumulator registers as bo_tower_h.h; what changes:
//   * a wave's 32*MT output channels x 128 positions (two boards) are 2*MT x 8 accumulator tiles of 16 x 16; a K-step is 32 input channels
//     of one tap: 2*MT weight fragments (buffer_load_dwordx4, AR = 6 steps ahead), 8 B operands (ds_read_b128, one step ahead),
//     16*MT MFMAs;
//   * weight layout per layer [tap 9][c_in/32][C/16][64 lanes][8 fp16]: lane l of a fragment holds W[16*tile + (l & 15)]
//     [32*group + 8*(l >> 4) + i][tap] (fused_net.pack_conv_weight_f16_t16); head weights as in bo_tower_h.h (32x32x16 tiles);
//   * LDS images with bo_tower_s16.h's chunk swizzle (j ^ 2*(column & 7) on the low four bits of the chunk index).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_tower_h.h"
#include "bo_tower_s16.h"

template <int PH>
__device__ inline int bo_sw16_addr_c(int cell, int chunk) { return cell * PH + ((chunk & ~15) << 3) + (((chunk ^ bo_sw16(cell)) & 15) << 3); }

template <int C, int MT, int AR = 6>
__global__ void __launch_bounds__(256)
bo_k_tower_h16(const float *__restrict__ x, const bo_h8 *__restrict__ wts, const float *__restrict__ params,
               const bo_tower_layer *__restrict__ layers, int n_layers, int B, bo_tower_head_h head) {
    constexpr int NW = 4, NT = 256, PH = C, CELLS = 100, IMGH = CELLS * PH, CIN0 = 120, HPW = 16 / NW, OT = 2 * MT;
    constexpr int UNR = 12;  // K-steps per unrolled body: three groups of four (a group = 128 input channels of one tap)
    static_assert(C == 32 * MT * NW && (C == 128 || C == 256) && UNR % AR == 0, "four waves of MT 32-channel tiles");
    __shared__ __attribute__((aligned(16))) _Float16 X[2 * IMGH];  // [board][cell][PH]
    __shared__ __attribute__((aligned(16))) float pooled[2][C];
    __shared__ float hid[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kb = lane >> 4;   // 16x16x32 operands
    const int n32 = lane & 31, kg = lane >> 5;   // 32x32x16 operands (head convolutions), SE gate ownership
    const int col0 = (n16 & 7) + 1;
    const int cell0 = ((n16 >> 3) + 1) * 10 + col0;            // padded cell of position n16; position n16 + 16*pt is cell0 + 20*pt
    const int cell32 = ((n32 >> 3) + 1) * 10 + (n32 & 7) + 1;  // of position n32; position n32 + 32 is cell32 + 40

    for (int i = tid; i < 2 * IMGH / 8; i += NT) reinterpret_cast<bo_h8 *>(X)[i] = bo_h8{0, 0, 0, 0, 0, 0, 0, 0};

    bo_f32x4v acc[OT][8];      // [channel tile][board*4 + position tile]: rows = channels 32*MT*wave + 16*ot + 4*kb + r, col = position 16*pt + n16
    bo_h8 a[AR][OT];           // weight fragments of AR consecutive K-steps
    bo_h8 bq[2][8];            // B operands of two consecutive K-steps
    bo_h4 skip[OT][8];         // block input at this lane's (channels, positions), packed like the LDS writes
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bo_h8 *>(wts), 0, 0x7fffffff, 0x00020000);
    const int wvoff = ((wave * OT) * 64 + lane) * 16;
    auto load_a = [&](int j, int w_off8, int step) {
        typedef int bo_i32x4_t __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane((w_off8 + step * (C / 16) * 64) * 16);
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const bo_i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + ot * 64 * 16, soff, 0);
            a[j][ot] = __builtin_bit_cast(bo_h8, v);
        }
    };
    auto read_b = [&](bo_h8(&b)[8], int addr) {  // addr: halves, board 0, position tile 0
        const _Float16 *p = X + addr;
#pragma unroll
        for (int t = 0; t < 8; t++) b[t] = *reinterpret_cast<const bo_h8 *>(p + (t >> 2) * IMGH + (t & 3) * 20 * PH);
    };
#define BO_H16_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#pragma unroll
    for (int j = 0; j < AR; j++) load_a(j, layers[0].w_off4, j);
    const int npairs = (B + 1) >> 1;
    for (int pb = blockIdx.x; pb < npairs; pb += gridDim.x) {
        const int b0 = 2 * pb;
        // ---- stage the 120 input planes of both boards (fp32 NCHW) as fp16 channels-last; channels >= 120 stay zero ----
        __syncthreads();
        for (int i = tid; i < 2 * 128 * 16; i += NT) {
            const int bb = i >> 11, rem = i & 2047, ic = rem >> 4, q = rem & 15;
            bo_f32x4 t = {0, 0, 0, 0};
            if (ic < CIN0 && b0 + bb < B) t = reinterpret_cast<const bo_f32x4 *>(x + (size_t)(b0 + bb) * CIN0 * 64)[rem];
            const int cell = ((q >> 1) + 1) * 10 + (q & 1) * 4 + 1;
#pragma unroll
            for (int e = 0; e < 4; e++) X[bb * IMGH + bo_sw16_addr_c<PH>(cell + e, ic >> 3) + (ic & 7)] = (_Float16)t[e];
        }
        __syncthreads();
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = layers[l];
            const bo_tower_layer Ln = layers[l + 1 < n_layers ? l + 1 : 0];
            const int T32 = L.t4 >> 1;       // K-steps of 32 channels (L.t4 counts steps of 16)
            const int gpt = L.t4 / 72;       // groups of 128 input channels per tap: 1 (the padded input conv, 128 filters) or 2
            float bv[OT][4];
#pragma unroll
            for (int ot = 0; ot < OT; ot++)
#pragma unroll
                for (int r = 0; r < 4; r++) bv[ot][r] = params[L.bias_off + wave * 32 * MT + ot * 16 + kb * 4 + r];
#pragma unroll
            for (int ot = 0; ot < OT; ot++)
#pragma unroll
                for (int t = 0; t < 8; t++) acc[ot][t] = bo_f32x4v{0, 0, 0, 0};
            // group g (four K-steps): tap g / gpt, input channels 128*(g % gpt) ..; its B address without the channel group of the step
            auto g_base = [&](int g) {
                const int tapn = g / gpt, hs = g - tapn * gpt, dx = tapn % 3 - 1;
                return (cell0 + (tapn / 3 - 1) * 10 + dx) * PH + hs * 128 + ((kb ^ (2 * ((col0 + dx) & 7))) << 3);
            };
            const int ngroups = T32 >> 2;
            int basec = g_base(0), basen = basec;
            read_b(bq[0], basec);
            for (int s0 = 0; s0 < T32; s0 += UNR) {
#pragma unroll
                for (int j = 0; j < UNR; j++) {
                    if ((j & 3) == 0) {
                        basec = (j == 0 && s0 == 0) ? basec : basen;
                        const int gn = ((s0 + j) >> 2) + 1;  // the group after this one (a harmless re-read at the layer's end)
                        basen = g_base(gn < ngroups ? gn : ngroups - 1);
                    }
                    const bo_h8(&bc)[8] = bq[j & 1];
                    // the step after this one: the next channel group of this group (chunk index ^ 4*cg on bits 2..3), or the next group's first
                    read_b(bq[(j + 1) & 1], (j & 3) < 3 ? (basec ^ ((((j & 3) + 1) * 4) << 3)) : basen);
#pragma unroll
                    for (int ot = 0; ot < OT; ot++)
#pragma unroll
                        for (int t = 0; t < 8; t++) acc[ot][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j % AR][ot], bc[t], acc[ot][t], 0, 0, 0);
                    const int sn = s0 + j + AR;  // this set's next owner: AR steps ahead, maybe in the next layer
                    load_a(j % AR, sn < T32 ? L.w_off4 : Ln.w_off4, sn < T32 ? sn : sn - T32);
#pragma unroll
                    for (int t = 0; t < 8; t++) { BO_H16_SGB(0x008, MT); BO_H16_SGB(0x100, 1); }
#pragma unroll
                    for (int ot = 0; ot < OT; ot++) { BO_H16_SGB(0x008, 4); BO_H16_SGB(0x020, 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();  // every wave has read the layer input: the output may overwrite it

            // ---- epilogue: a lane holds, per (channel tile, board, position tile), 4 consecutive channels of one position ----
            float gate[OT][2][4];
            if (L.kind == 3) {
                // SE gate (network.py:33-45) for both boards.  Wave w owns hidden units w, w + 4, ...; lane (n32, kg) owns the gates of
                // channels 32*(MT*wave + mt) + n32 of board kg.
                const float *w1 = params + L.se_w1_off, *w2 = params + L.se_w2_off;
                const bool have4 = lane < C / 4;
                float w2r[MT][16];
                bo_f32x4 w1r[HPW];
#pragma unroll
                for (int u = 0; u < HPW; u++)
                    w1r[u] = (have4 && wave + u * NW < L.hidden) ? reinterpret_cast<const bo_f32x4 *>(w1 + (size_t)(wave + u * NW) * C)[lane] : bo_f32x4{0, 0, 0, 0};
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int h = 0; h < 16; h++) w2r[mt][h] = h < L.hidden ? w2[((wave * MT + mt) * 32 + n32) * L.hidden + h] : 0.0f;
#pragma unroll
                for (int ot = 0; ot < OT; ot++)
#pragma unroll
                    for (int bb = 0; bb < 2; bb++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float s = bo_row_sum((acc[ot][4 * bb][r] + acc[ot][4 * bb + 1][r]) + (acc[ot][4 * bb + 2][r] + acc[ot][4 * bb + 3][r]));
                            if (n16 == 0) pooled[bb][wave * 32 * MT + ot * 16 + kb * 4 + r] = s * (1.0f / 64.0f) + bv[ot][r];
                        }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < HPW; u++)
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) {
                        const bo_f32x4 m = have4 ? reinterpret_cast<const bo_f32x4 *>(pooled[bb])[lane] : bo_f32x4{0, 0, 0, 0};
                        float p = (w1r[u][0] * m[0] + w1r[u][1] * m[1]) + (w1r[u][2] * m[2] + w1r[u][3] * m[3]);
                        p = bo_wave_sum63(p);
                        if (lane == 63 && wave + u * NW < L.hidden) hid[bb][wave + u * NW] = fmaxf(p, 0.0f);
                    }
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    float g = 0.0f;
#pragma unroll
                    for (int h = 0; h < 16; h++)
                        if (h < L.hidden) g += w2r[mt][h] * hid[kg][h];
                    g = 1.0f / (1.0f + expf(-g));
#pragma unroll
                    for (int o2 = 0; o2 < 2; o2++)
#pragma unroll
                        for (int bb = 0; bb < 2; bb++)
#pragma unroll
                            for (int r = 0; r < 4; r++) gate[2 * mt + o2][bb][r] = __shfl(g, o2 * 16 + kb * 4 + r + 32 * bb);
                }
            }
#pragma unroll
            for (int ot = 0; ot < OT; ot++)
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int ch0 = wave * 32 * MT + ot * 16 + kb * 4;
                    bo_h4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float v = acc[ot][t][r] + bv[ot][r];
                        if (L.kind == 3) v = v * gate[ot][t >> 2][r];
                        if (L.kind >= 2) v += (float)skip[ot][t][r];
                        o[r] = (_Float16)fmaxf(v, 0.0f);
                    }
                    if (L.kind != 1) skip[ot][t] = o;
                    *reinterpret_cast<bo_h4 *>(X + (t >> 2) * IMGH + bo_sw16_addr_c<PH>(cell0 + 20 * (t & 3), ch0 >> 3) + (ch0 & 7)) = o;
                }
            __syncthreads();
        }
        // ---- the two 1x1 head convolutions + ReLU on the tower output in X: one 32x32 job per (32 head channels, board, half) ----
        if (head.channels > 0) {
            const int mts = (head.channels + 31) >> 5;
            for (int job = wave; job < mts * 4; job += NW) {
                const int mt = job >> 2, t = job & 3, bb = t >> 1;
                bo_f32x16 hacc;
#pragma unroll
                for (int r = 0; r < 16; r++) hacc[r] = 0.0f;
#pragma unroll 4
                for (int st = 0; st < C / 16; st++) {
                    const bo_h8 aw = wts[(size_t)head.w_off8 + ((size_t)mt * (C / 16) + st) * 64 + lane];
                    const _Float16 *xb = X + bb * IMGH + bo_sw16_addr_c<PH>(cell32 + 40 * (t & 1), 2 * st + kg);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, *reinterpret_cast<const bo_h8 *>(xb), hacc, 0, 0, 0);
                }
                if (b0 + bb < B) {
                    const int sq = 32 * (t & 1) + n32;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int oc = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (oc < head.channels) {
                            const _Float16 v = (_Float16)fmaxf(hacc[r] + params[head.b_off + oc], 0.0f);
                            if (oc < head.split) head.out_a[((size_t)(b0 + bb) * head.split + oc) * 64 + sq] = v;
                            else head.out_b[((size_t)(b0 + bb) * (head.channels - head.split) + (oc - head.split)) * 64 + sq] = v;
                        }
                    }
                }
            }
        }
    }
}
#endif
