// betaone_amd/csrc/bo_tower_s.h -- the residual tower in float32 on the fp16 matrix pipe: every float32 operand is carried as a
// PAIR of fp16 values (hi = RN16(v), lo = RN16(v - hi): 22 significant bits) and every product as three
// v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulation; the dropped lo*lo term is 2^-22 of the product).
// Same contract as bo_tower.h (one workgroup keeps one board in LDS for the whole tower of
// /root/reference/network.py:48-118,167-195, BatchNorm folded; float32 planes in, float32 head planes / tower output out).
//
// Why: the fp32 matrix pipe is 1/16 of the fp16 one (157 vs ~2500 TFLOP/s dense).  bo_tower_wg.h sits at 0.80 of the fp32
// peak with Winograd; three fp16 MFMAs per product in the DIRECT form cost 3 x 2.25 / 16 = 0.42 of the Winograd form's
// matrix time, and the direct form streams 9/16 of the Winograd form's weight bytes (the stream from L2, ~56 B/clk/CU, is
// what bounds a split-precision Winograd layer).
//   * activations in LDS channels-last as two fp16 images (hi, lo) of [padded 10x10 cell][C]: the B operand of a lane
//     (8 consecutive channels of one cell) is one ds_read_b128 per image, conflict-free through the chunk swizzle of
//     bo_tower_h.h (bo_sw); ONE image pair per board, the layer's output overwrites its input after a barrier; the skip
//     connection stays in registers as float32;
//   * weights pre-split on the host, scaled per layer by a power of two so that the largest |w| sits just below 2^15 (a lo
//     half is then a normal fp16 number for every weight above 2^-18 of the largest); the epilogue multiplies the
//     accumulator by the inverse (exact) in the fma that adds the bias;
//   * M = 32*MT output channels per wave (4 waves, one per SIMD), N = 64 positions (two 32-wide tiles), K = 9 taps x C in
//     steps of 16 channels; A fragments [step][oc/32][hi|lo][lane][8 fp16] = one buffer_load_dwordx4 each, reloaded AR steps
//     ahead into the register set its own MFMAs just released (AR = 12 for the 128-filter instance; 4 for 256 filters, whose
//     two tiles per wave double every register set: with 8 the instance spilled 41 registers).
// Numerics (tests/test_engine_gpu.py: test_fused_epilogue_net_matches_plain_net): within 1e-5 of the float32 layer-by-layer net;
// G1 fixtures (3 sizes, the reference-default 15+5 x 256 among them) within 1e-4.
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include <type_traits>
#include "bo_tower.h"
#include "bo_tower_h.h"

struct bo_tower_head_s {
    int channels = 0, split = 0, w_off8 = 0, b_off = 0;  // head weights: [mt][step C/16][hi|lo][lane][8 fp16] at bo_h8 offset w_off8 in wts;
    float *out_a = nullptr, *out_b = nullptr;            // params[b_off + channels] = their inverse scale
    int *overflow = nullptr;                              // set to 1 when an activation had to be saturated to the fp16 range (the result is then wrong)
    unsigned long long *timing = nullptr;                 // bo_nn_tower_forward_timed: [seq | arrivals | start[BO_TOWER_TIMING_CAP] | end[..]] of the launches
                                                          // made with this buffer, in the device's constant-rate clock (wall_clock64)
};
#define BO_TOWER_TIMING_CAP 4096

// hi / lo halves of 4 float32 values (saturating: |v| beyond the fp16 range would turn into inf - inf)
__device__ inline void bo_split4(const float (&v)[4], bo_h4 &hi, bo_h4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float c = fminf(fmaxf(v[e], -65504.0f), 65504.0f);
        hi[e] = (_Float16)c;
        lo[e] = (_Float16)(c - (float)hi[e]);
    }
}

// the same for values that are >= 0 already (behind a ReLU); `sat` collects whether anything was cut
__device__ inline void bo_split4_pos(const float (&v)[4], bo_h4 &hi, bo_h4 &lo, bool &sat) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float c = fminf(v[e], 65504.0f);
        sat = sat || !(v[e] <= 65504.0f);  // (also a NaN)
        hi[e] = (_Float16)c;
        lo[e] = (_Float16)__builtin_fmaf((float)hi[e], -1.0f, c);  // = c - hi, one rounding (v_fma_mix: the fp16 operand is widened on read)
    }
}

// BD = how many K-steps ahead of its MFMAs a B operand is read from LDS (1: two register sets; 2, 3: four);
// AR = weight-fragment sets = how many K-steps ahead a weight fragment is requested (4 or 8, or 12 with the loop unrolled 24-fold)
template <int C, int MT, int BD = 1, int AR = 8>
__global__ void __launch_bounds__(256)
bo_k_tower_s(const float *__restrict__ x, const bo_h8 *__restrict__ wts, const float *__restrict__ params,
             const bo_tower_layer *__restrict__ layers, int n_layers, float *__restrict__ y, int B, bo_tower_head_s head) {
    constexpr int NW = 4, NT = 256, PH = C, CELLS = 100, IMGH = CELLS * PH, CIN0 = 120, HPW = 16 / NW;
    static_assert(C == 32 * MT * NW, "four waves of MT 32-channel tiles");
    __shared__ __attribute__((aligned(16))) _Float16 X[2 * IMGH];  // [hi | lo][cell][PH]
    __shared__ __attribute__((aligned(16))) float pooled[C];
    __shared__ float hid[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = lane >> 5, n = lane & 31;
    const int cell0 = ((n >> 3) + 1) * 10 + (n & 7) + 1;  // padded cell of position n; position n + 32 is cell0 + 40

    // launch timing (bench.py's live roofline leg; works inside captured graphs, where no event pair can sit between two nodes): the
    // first workgroup notes when it starts, the workgroup that finishes last notes when, in slot seq % CAP of the caller's buffer;
    // seq only moves when a launch has ended, so every workgroup of a launch reads the same value (one launch per buffer at a time).
    unsigned long long tseq = 0;
    if (head.timing) {
        tseq = __hip_atomic_load(head.timing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0 && tid == 0) head.timing[2 + tseq % BO_TOWER_TIMING_CAP] = wall_clock64();
    }
    for (int i = tid; i < 2 * IMGH / 8; i += NT) reinterpret_cast<bo_h8 *>(X)[i] = bo_h8{0, 0, 0, 0, 0, 0, 0, 0};

    bo_f32x16 acc[MT][2];      // [tile][position half]: rows = channels 32*(MT*wave + tile) + (r&3) + 8*(r>>2) + 4*kg, col = position n + 32*half
    constexpr int UNR = (AR == 8 || AR == 4) ? 8 : 24;
    static_assert((AR == 4 || AR == 6 || AR == 8 || AR == 12) && BD >= 1 && BD <= 3, "ring sizes the unrolled loop can index statically");
    bo_h8 a[AR][MT][2];        // A fragments (hi, lo) of AR consecutive K-steps
    constexpr int BM = BD == 1 ? 1 : 3;
    bo_h8 bq[BM + 1][2][2];    // B operands of consecutive K-steps: [set][position half][hi | lo]
    float skip[MT][2][16];     // block input at this lane's (channels, positions), float32
    const int sw0 = bo_sw(cell0);  // (position n + 32 sits 4 rows further down: the same swizzle)
    // B operand address (in halves) of K-step j of a group of 8: tap cell offset `tc`, first channel group cg0 (a multiple of 8)
    auto b_base = [&](int tc, int cg0) { return (cell0 + tc) * PH + 16 * cg0 + (((kg ^ bo_sw(cell0 + tc)) & 15) << 3); };
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bo_h8 *>(wts), 0, 0x7fffffff, 0x00020000);
    const int wvoff = ((wave * MT * 2) * 64 + lane) * 16;
    auto load_a = [&](int j, int w_off8, int step) {
        typedef int bo_i32x4_t __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane((w_off8 + step * (C / 32) * 2 * 64) * 16);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int hl = 0; hl < 2; hl++) {
                const bo_i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + (mt * 2 + hl) * 64 * 16, soff, 0);
                a[j][mt][hl] = __builtin_bit_cast(bo_h8, v);
            }
    };
    auto read_b = [&](bo_h8(&b)[2][2], int base, int j) {  // K-step j of the group whose b_base() is `base`
        const _Float16 *p = X + (base ^ (j << 4));
        b[0][0] = *reinterpret_cast<const bo_h8 *>(p);
        b[0][1] = *reinterpret_cast<const bo_h8 *>(p + IMGH);
        b[1][0] = *reinterpret_cast<const bo_h8 *>(p + 40 * PH);
        b[1][1] = *reinterpret_cast<const bo_h8 *>(p + IMGH + 40 * PH);
    };
#define BO_S_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#pragma unroll
    for (int j = 0; j < AR; j++) load_a(j, layers[0].w_off4, j);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- stage the 120 input planes (float32 NCHW) as fp16 pairs, channels-last; channels >= 120 stay zero ----
        __syncthreads();
        for (int i = tid; i < 128 * 16; i += NT) {
            const int ic = i >> 4, q = i & 15;
            bo_f32x4 t = {0, 0, 0, 0};
            if (ic < CIN0) t = reinterpret_cast<const bo_f32x4 *>(x + (size_t)b * CIN0 * 64)[i];
            const float tv[4] = {t[0], t[1], t[2], t[3]};
            bo_h4 hi, lo;
            bo_split4(tv, hi, lo);
            const int cell = ((q >> 1) + 1) * 10 + (q & 1) * 4 + 1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                _Float16 *dst = X + bo_sw_addr<PH>(cell + e, ic >> 3) + (ic & 7);
                dst[0] = hi[e]; dst[IMGH] = lo[e];
            }
        }
        __syncthreads();
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = layers[l];
            const bo_tower_layer Ln = layers[l + 1 < n_layers ? l + 1 : 0];
            const int ncg = L.t4 / 9;  // channel groups of 16 per tap (L.t4 = K-steps of the layer, a multiple of 8)
            const float wscale = params[L.bias_off + C];  // 2^-T: the layer's weights were stored multiplied by 2^T
            float bv[MT][16];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) bv[mt][r] = params[L.bias_off + (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int t = 0; t < 2; t++)
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
                    const bo_h8(&bc)[2][2] = bq[j & BM];
                    read_b(bq[(j + BD) & BM], (j & 7) + BD < 8 ? basec : basen, (j + BD) & 7);
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) {  // the small terms first
                            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j % AR][mt][1], bc[t][0], acc[mt][t], 0, 0, 0);
                            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j % AR][mt][0], bc[t][1], acc[mt][t], 0, 0, 0);
                            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j % AR][mt][0], bc[t][0], acc[mt][t], 0, 0, 0);
                        }
                    const int sn = s0 + j + AR;  // this set's next owner: AR steps ahead, maybe in the next layer
                    load_a(j % AR, sn < L.t4 ? L.w_off4 : Ln.w_off4, sn < L.t4 ? sn : sn - L.t4);
                    // every LDS read and weight load in the shadow of a different MFMA
#pragma unroll
                    for (int t = 0; t < 4; t++) { BO_S_SGB(0x008, 1); BO_S_SGB(0x100, 1); }
                    if (MT > 1) BO_S_SGB(0x008, 4 * MT - 4);
#pragma unroll
                    for (int i = 0; i < 2 * MT; i++) { BO_S_SGB(0x008, 1); BO_S_SGB(0x020, 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();  // every wave has read the layer input: the output may overwrite it

            // ---- epilogue: rows (r&3) + 8*(r>>2) + 4*kg of a tile are 4 consecutive channels per r>>2 ----
            float gate[MT][16];
            if (L.kind == 3) {
                // SE gate (network.py:33-45).  Wave w owns hidden units w, w + 4, ...; lane n owns the gate of channel 32*(MT*wave + tile) + n.
                // Weights are requested first, reductions run on the VALU (DPP).
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
                    for (int r = 0; r < 16; r++) {
                        const float s = bo_half_sum(acc[mt][0][r] + acc[mt][1][r]);
                        if (n == 16) pooled[(wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] = s * (wscale * (1.0f / 64.0f)) + bv[mt][r];
                    }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < HPW; u++) {
                    const bo_f32x4 m = have4 ? reinterpret_cast<const bo_f32x4 *>(pooled)[lane] : bo_f32x4{0, 0, 0, 0};
                    float p = (w1r[u][0] * m[0] + w1r[u][1] * m[1]) + (w1r[u][2] * m[2] + w1r[u][3] * m[3]);
                    p = bo_wave_sum63(p);
                    if (lane == 63 && wave + u * NW < L.hidden) hid[wave + u * NW] = fmaxf(p, 0.0f);
                }
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    float g = 0.0f;
#pragma unroll
                    for (int h = 0; h < 16; h++)
                        if (h < L.hidden) g += w2r[mt][h] * hid[h];
                    g = 1.0f / (1.0f + expf(-g));
#pragma unroll
                    for (int r = 0; r < 16; r++) gate[mt][r] = __shfl(g, (r & 3) + 8 * (r >> 2) + 4 * kg);
                }
            }
            // (one straight-line copy per layer kind: with the kind tested per element the compiler emitted ~160 selects per layer)
            auto write_back = [&](auto kind_c, auto y_c) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr bool TO_Y = decltype(y_c)::value;
                bool sat = false;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int ch0 = (wave * MT + mt) * 32 + 4 * kg;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            _Float16 *cellp = X + 40 * t * PH + bo_sw_addr<PH>(cell0, (wave * MT + mt) * 4 + q, sw0) + 4 * kg;
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int r = 4 * q + e;
                                float v = __builtin_fmaf(acc[mt][t][r], wscale, bv[mt][r]);  // (wscale is a power of two: the product is exact either way)
                                if (KIND == 3) v = v * gate[mt][r];
                                if (KIND >= 2) v += skip[mt][t][r];
                                o[e] = fmaxf(v, 0.0f);
                                if (KIND != 1) skip[mt][t][r] = o[e];
                            }
                            bo_h4 hi, lo;
                            bo_split4_pos(o, hi, lo, sat);
                            *reinterpret_cast<bo_h4 *>(cellp) = hi;
                            *reinterpret_cast<bo_h4 *>(cellp + IMGH) = lo;
                            if (TO_Y) {
                                float *g2 = y + ((size_t)b * C + ch0 + 8 * q) * 64 + n + 32 * t;
#pragma unroll
                                for (int e = 0; e < 4; e++) g2[e * 64] = o[e];
                            }
                        }
                    }
                // an activation beyond the fp16 range cannot be carried as a (hi, lo) pair: the tower's result is wrong from here on --
                // say so (bo_nn_tower_status) instead of returning it silently; such a net needs the fp32-pipe tower
                if (__ballot(sat) != 0ull && lane == 0 && head.overflow) atomicOr(head.overflow, 1);
            };
            using std::integral_constant;
            if (L.last && y) {
                if (L.kind == 3) write_back(integral_constant<int, 3>{}, std::true_type{});
                else write_back(integral_constant<int, 2>{}, std::true_type{});
            } else if (L.kind == 0) write_back(integral_constant<int, 0>{}, std::false_type{});
            else if (L.kind == 1) write_back(integral_constant<int, 1>{}, std::false_type{});
            else if (L.kind == 2) write_back(integral_constant<int, 2>{}, std::false_type{});
            else write_back(integral_constant<int, 3>{}, std::false_type{});
            __syncthreads();
        }
        // ---- the two 1x1 head convolutions + ReLU on the tower output in X: one 32x32 job per (32 head channels, position half) ----
        if (head.channels > 0) {
            const int mts = (head.channels + 31) >> 5;
            const float hscale = params[head.b_off + head.channels];
            for (int job = wave; job < mts * 2; job += NW) {
                const int mt = job >> 1, t = job & 1;
                bo_f32x16 hacc;
#pragma unroll
                for (int r = 0; r < 16; r++) hacc[r] = 0.0f;
#pragma unroll 4
                for (int st = 0; st < C / 16; st++) {
                    const bo_h8 *wp = wts + (size_t)head.w_off8 + (((size_t)mt * (C / 16) + st) * 2) * 64 + lane;
                    const bo_h8 ah = wp[0], al = wp[64];
                    const _Float16 *xb = X + t * 40 * PH + bo_sw_addr<PH>(cell0, 2 * st + kg, sw0);
                    const bo_h8 bh = *reinterpret_cast<const bo_h8 *>(xb), bl = *reinterpret_cast<const bo_h8 *>(xb + IMGH);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hacc, 0, 0, 0);
                }
                const int sq = 32 * t + n;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int oc = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (oc < head.channels) {
                        const float v = fmaxf(hacc[r] * hscale + params[head.b_off + oc], 0.0f);
                        if (oc < head.split) head.out_a[((size_t)b * head.split + oc) * 64 + sq] = v;
                        else head.out_b[((size_t)b * (head.channels - head.split) + (oc - head.split)) * 64 + sq] = v;
                    }
                }
            }
        }
    }
    if (head.timing) {
        __syncthreads();
        if (tid == 0) {
            const unsigned long long arrived = __hip_atomic_fetch_add(head.timing + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived + 1 == gridDim.x) {  // the launch's last workgroup
                head.timing[2 + BO_TOWER_TIMING_CAP + tseq % BO_TOWER_TIMING_CAP] = wall_clock64();
                __hip_atomic_store(head.timing + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(head.timing, tseq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
#endif