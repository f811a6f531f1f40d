// betaone_amd/csrc/bo_tower_h.h -- the residual tower for 256-filter nets with an fp16 evaluate stage (BASELINE.json
// configs[4]: 20-block x 256 net, "fp16 net eval") as one persistent kernel on v_mfma_f32_32x32x16_f16 (gfx950).
//
// Same contract as bo_tower.h (/root/reference/network.py:48-118,167-195, BatchNorm folded): fp16 weights and
// activations, fp32 accumulation, like the network in torch.float16.  What changes against the fp32 kernels:
//   * fp16 MFMA is 16x the fp32 rate, so a 3x3 layer of one board costs 7.7 us of MFMA time while its 1.18 MB of
//     weights need ~17 us to stream from L2 to a CU: a workgroup therefore keeps TWO boards (512 games per GPU =
//     2 boards per CU) and every weight fragment feeds 4 MFMAs (2 boards x 2 position halves);
//   * activations live in LDS channels-last, [board][padded 10x10 cell][C] fp16, the 16-byte chunks of a cell swizzled
//     (bo_sw below: the B operand of a lane, 8 consecutive channels of one cell = one ds_read_b128, is then
//     bank-conflict free for every tap; the padded pitch of C + 8 used before was 3-way conflicted).  ONE image
//     per board: each wave holds its 32 output channels x all 128 positions in 64 accumulator registers, so after a
//     barrier the layer's output overwrites its input in place; the skip connection stays in registers (packed fp16);
//   * direct (not Winograd) convolution: M = 32 output channels per wave (8 waves), N = 128 positions, K = 9 taps x C
//     in steps of 16 channels; A fragments [step][oc/32][lane][8 fp16] = one global_load_dwordx4, reloaded 8 steps
//     ahead into the register set its own MFMAs just released (64 KB per CU in flight).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_tower.h"

typedef _Float16 bo_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 bo_h4 __attribute__((ext_vector_type(4)));

// LDS images [padded 10x10 cell][C fp16 channels]: a lane's B operand is one 16-byte chunk (8 channels) of a cell, and a
// ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} of each wave half,
// MI355X_MICROARCH.md section LDS) whose 16 cells must sit on 16 different bank quads.  With a linear cell pitch two cells
// 16 apart (rows r and r+2 of a 10-wide image overlap that way) always collide: the pitch of C + 8 halves used before cost
// three LDS cycles per group instead of one, and the four waves' reads then took as long as the MFMAs they feed.  So chunk
// j of a cell is stored at chunk position j ^ bo_sw(cell) (low four bits): for every tap and both lane groups the 16 cells
// of a group get 16 different positions (checked exhaustively, tests/test_abi.py).
__device__ __host__ inline int bo_sw(int cell) { return ((cell % 10) + 8 * (cell / 10)) & 15; }
template <int PH>
__device__ inline int bo_sw_addr(int cell, int chunk, int sw) { return cell * PH + ((chunk & ~15) << 3) + (((chunk ^ sw) & 15) << 3); }
template <int PH>
__device__ inline int bo_sw_addr(int cell, int chunk) { return bo_sw_addr<PH>(cell, chunk, bo_sw(cell)); }

struct bo_tower_head_h {
    int channels = 0, split = 0, w_off8 = 0, b_off = 0;  // head weights: [mt 2][step C/16][lane][8 fp16] at bo_h8 offset w_off8 in wts
    _Float16 *out_a = nullptr, *out_b = nullptr;
};

// LAB (scripts/conv_lab.hip only): 0 = the kernel; 1 = no weight loads; 2 = no B operand reads; 4 = no epilogue;
// 5 = 1 + 2 + 4; 6 = 5 without barriers
// MT = 32-channel output tiles per wave: C/(32*MT) waves per workgroup.  MT = 2 (256 filters: 4 waves, one per SIMD, 64
// channels x 128 positions = 128 accumulator registers) reads every B operand for two MFMAs and halves the LDS traffic.
// BD = how many K-steps ahead of its MFMAs a B operand is read from LDS (1: two register sets; 2, 3: four); AR = weight-fragment
// sets = how many K-steps ahead a fragment is requested (8, or 12 with the loop unrolled 24-fold) -- see bo_tower_s.h
template <int C, int MT, int LAB = 0, int BD = 1, int AR = 8>
__global__ void __launch_bounds__(C * 2 / MT)
bo_k_tower_h(const float *__restrict__ x, const bo_h8 *__restrict__ wts, const float *__restrict__ params,
             const bo_tower_layer *__restrict__ layers, int n_layers, int B, bo_tower_head_h head) {
    constexpr int NW = C / (32 * MT), NT = NW * 64, PH = C, CELLS = 100, IMGH = CELLS * PH, CIN0 = 120, HPW = (16 + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) _Float16 X[2 * IMGH];
    __shared__ __attribute__((aligned(16))) float pooled[2][C];
    __shared__ float hid[2][16];
    static_assert(C == 128 || C == 256, "one float4 of a W1 row per lane");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = lane >> 5, n = lane & 31;
    const int cell0 = ((n >> 3) + 1) * 10 + (n & 7) + 1;  // padded cell of position n; position n + 32 is cell0 + 40

    for (int i = tid; i < 2 * IMGH / 8; i += NT) reinterpret_cast<bo_h8 *>(X)[i] = bo_h8{0, 0, 0, 0, 0, 0, 0, 0};

    bo_f32x16 acc[MT][4];      // [tile][board*2 + half]: rows = channels 32*(MT*wave + tile) + (r&3) + 8*(r>>2) + 4*kg, col = position n
    constexpr int UNR = AR == 8 ? 8 : 24, BM = BD == 1 ? 1 : 3;
    static_assert((AR == 8 || AR == 12) && BD >= 1 && BD <= 3, "ring sizes the unrolled loop can index statically");
    bo_h8 a[AR][MT];           // A fragments of AR consecutive K-steps
    bo_h8 bq[BM + 1][4];       // B operands of consecutive K-steps
    bo_h4 skip[MT][4][4];      // block input at this lane's (positions, channels), packed like the LDS writes
    const int sw0 = bo_sw(cell0);  // (position n + 32 sits 4 rows further down: the same swizzle)
    // B operand address (in halves) of K-step j of a group of 8: tap cell offset `tc`, first channel group cg0 (a multiple of 8)
    auto b_base = [&](int tc, int cg0) { return (cell0 + tc) * PH + 16 * cg0 + (((kg ^ bo_sw(cell0 + tc)) & 15) << 3); };
    // weight fragments through a buffer descriptor: per-thread offset in one VGPR, layer / K-step offset in scalar registers,
    // no vector address arithmetic per load (see bo_tower_wg.h)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bo_h8 *>(wts), 0, 0x7fffffff, 0x00020000);
    const int wvoff = ((wave * MT) * 64 + lane) * 16;
    auto load_a = [&](int j, int w_off8, int step) {
        if (LAB == 1 || LAB >= 5) return;
        typedef int bo_i32x4_t __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane((w_off8 + step * (C / 32) * 64) * 16);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const bo_i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + mt * 64 * 16, soff, 0);
            a[j][mt] = __builtin_bit_cast(bo_h8, v);
        }
    };
    auto read_b = [&](bo_h8(&b)[4], int base, int j) {  // K-step j of the group whose b_base() is `base`
        if (LAB == 2 || LAB >= 5) return;
        const _Float16 *p = X + (base ^ (j << 4));
        b[0] = *reinterpret_cast<const bo_h8 *>(p);
        b[1] = *reinterpret_cast<const bo_h8 *>(p + 40 * PH);
        b[2] = *reinterpret_cast<const bo_h8 *>(p + IMGH);
        b[3] = *reinterpret_cast<const bo_h8 *>(p + IMGH + 40 * PH);
    };
#define BO_H_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#pragma unroll
    for (int j = 0; j < AR; j++) load_a(j, layers[0].w_off4, j);
    const int npairs = (B + 1) >> 1;
    for (int pb = blockIdx.x; pb < npairs; pb += gridDim.x) {
        const int b0 = 2 * pb;
        // ---- stage the 120 input planes of both boards (fp32 NCHW) as fp16 channels-last; channels >= 120 stay zero ----
        __syncthreads();
        for (int i = tid; i < 2 * 128 * 16; i += NT) {  // channels 120..127 of the padded input conv are zeroed
            const int bb = i >> 11, rem = i & 2047, ic = rem >> 4, q = rem & 15;
            bo_f32x4 t = {0, 0, 0, 0};
            if (ic < CIN0 && b0 + bb < B) t = reinterpret_cast<const bo_f32x4 *>(x + (size_t)(b0 + bb) * CIN0 * 64)[rem];
            const int cell = ((q >> 1) + 1) * 10 + (q & 1) * 4 + 1;
#pragma unroll
            for (int e = 0; e < 4; e++) X[bb * IMGH + bo_sw_addr<PH>(cell + e, ic >> 3) + (ic & 7)] = (_Float16)t[e];
        }
        __syncthreads();
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = layers[l];
            const bo_tower_layer Ln = layers[l + 1 < n_layers ? l + 1 : 0];
            const int ncg = L.t4 / 9;  // channel groups of 16 per tap (L.t4 = K-steps of the layer, a multiple of 8)
            float bv[MT][16];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) bv[mt][r] = params[L.bias_off + (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[mt][t][r] = 0.0f;
            int basec = 0, basen = b_base(-11, 0);  // step 0: tap 0 = (-1, -1), channel group 0
#pragma unroll
            for (int j = 0; j < BD; j++) read_b(bq[j], basen, j);
            for (int s0 = 0; s0 < L.t4; s0 += UNR) {
#pragma unroll
                for (int j = 0; j < UNR; j++) {
                    if ((j & 7) == 0) {  // a group of 8 steps = 8 channel groups of one tap
                        basec = basen;
                        // the group after this one (the next 8 channel groups or the next tap; a harmless re-read at the layer's end)
                        const int s8 = s0 + j + 8, sc = s8 < L.t4 ? s8 : s0 + j, tapn = sc / ncg;
                        basen = b_base((tapn / 3 - 1) * 10 + (tapn % 3 - 1), sc - tapn * ncg);
                    }
                    const bo_h8(&bc)[4] = bq[j & BM];
                    read_b(bq[(j + BD) & BM], (j & 7) + BD < 8 ? basec : basen, (j + BD) & 7);
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j % AR][mt], bc[t], acc[mt][t], 0, 0, 0);
                    const int sn = s0 + j + AR;  // this set's next owner: AR steps ahead, maybe in the next layer
                    load_a(j % AR, sn < L.t4 ? L.w_off4 : Ln.w_off4, sn < L.t4 ? sn : sn - L.t4);
                    // every LDS read and weight load in the shadow of a different MFMA
#pragma unroll
                    for (int t = 0; t < 4; t++) { BO_H_SGB(0x008, MT); BO_H_SGB(0x100, 1); }
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) BO_H_SGB(0x020, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (LAB != 6) __syncthreads();  // every wave has read the layer input: the output may overwrite it

            // ---- epilogue: rows (r&3) + 8*(r>>2) + 4*kg of a tile are 4 consecutive channels per r>>2 ----
            float gate[MT][2][16];
            if (L.kind == 3) {
                // SE gate (network.py:33-45) for both boards.  Wave w owns hidden units w, w + NW, ...; lane (n, kg) owns the gates
                // of channels 32*(MT*wave + tile) + n of board kg.  Weights are requested first, reductions run on the VALU (DPP).
                const float *w1 = params + L.se_w1_off, *w2 = params + L.se_w2_off;
                const bool have4 = lane < C / 4;  // a W1 row is C/4 float4: one per lane (C = 256) or per lane of the first half (128)
                float w2r[MT][16];
                bo_f32x4 w1r[HPW];
#pragma unroll
                for (int u = 0; u < HPW; u++)
                    w1r[u] = (have4 && wave + u * NW < L.hidden) ? reinterpret_cast<const bo_f32x4 *>(w1 + (size_t)(wave + u * NW) * C)[lane] : bo_f32x4{0, 0, 0, 0};
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int h = 0; h < 16; h++) w2r[mt][h] = h < L.hidden ? w2[((wave * MT + mt) * 32 + n) * L.hidden + h] : 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int bb = 0; bb < 2; bb++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float s = bo_half_sum(acc[mt][2 * bb][r] + acc[mt][2 * bb + 1][r]);
                            if (n == 16) pooled[bb][(wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] = s * (1.0f / 64.0f) + bv[mt][r];
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
                    for (int bb = 0; bb < 2; bb++)
#pragma unroll
                        for (int r = 0; r < 16; r++) gate[mt][bb][r] = __shfl(g, (r & 3) + 8 * (r >> 2) + 4 * kg + 32 * bb);
                }
            }
            if (LAB >= 4) {
                if (acc[0][0][0] == 123.456f) X[tid] = (_Float16)(acc[0][1][1] + acc[MT - 1][2][2] + acc[MT - 1][3][3]);
            } else
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        _Float16 *cellp = X + (t >> 1) * IMGH + 40 * (t & 1) * PH + bo_sw_addr<PH>(cell0, (wave * MT + mt) * 4 + q, sw0) + 4 * kg;
                        bo_h4 o;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int r = 4 * q + e;
                            float v = acc[mt][t][r] + bv[mt][r];
                            if (L.kind == 3) v = v * gate[mt][t >> 1][r];
                            if (L.kind >= 2) v += (float)skip[mt][t][q][e];
                            o[e] = (_Float16)fmaxf(v, 0.0f);
                        }
                        if (L.kind != 1) skip[mt][t][q] = o;
                        *reinterpret_cast<bo_h4 *>(cellp) = o;
                    }
                }
            if (LAB != 6) __syncthreads();
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
                    const _Float16 *xb = X + bb * IMGH + (t & 1) * 40 * PH + bo_sw_addr<PH>(cell0, 2 * st + kg, sw0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, *reinterpret_cast<const bo_h8 *>(xb), hacc, 0, 0, 0);
                }
                if (b0 + bb < B) {
                    const int sq = 32 * (t & 1) + n;
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
