// betaone_amd/csrc/bo_tower_s16.h -- bo_k_tower_s (bo_tower_s.h: the residual tower in float32 on the fp16 matrix pipe, every operand a
// (hi, lo) pair of fp16 values, three MFMAs per product) re-tiled for v_mfma_f32_16x16x32_f16.  128 filters.
//
// Why a second tiling of the same arithmetic (profiles/r05_tower_bound.md): one board per workgroup makes every CU stream every layer's
// 590 KB of weight pairs through its own port to L2, and that port delivers ~27 bytes per CLOCK -- a launch alone runs at exactly that
// rate (186 us).  Four cohorts' launches side by side take 218 us each because the chip lowers its clock under the matrix load
// (2.39 -> ~2.04 GHz).  With the whole chip multiplying random fp16 data, loops of 16x16x32 tiles hold 2.10 GHz where 32x32x16 tiles hold
// 1.93 (scripts/mfma_shape_lab.hip; MI355X_MICROARCH.md, DVFS give-back item 7), at 4 % fewer cycles for the same flops.  Same operand
// bytes, same accumulator registers, same results (a different summation order inside an MFMA: tested to the same tolerances).
//
// Against bo_tower_s.h:
//   * a wave still owns 32 output channels x 64 positions, now as 2 x 4 accumulator tiles of 16 x 16; a K-step is 32 input channels of
//     one tap (36 per layer): 4 weight fragments (2 channel tiles x (hi, lo); one buffer_load_dwordx4 each, requested AR = 6 steps = 96 KB
//     per CU ahead), 8 B operands (4 position tiles x (hi, lo); ds_read_b128, read one step ahead), 24 MFMAs;
//   * weight layout per layer [tap 9][c_in/32][C/16][hi | lo][64 lanes][8 fp16]: lane l of a fragment holds W[16*tile + (l & 15)]
//     [32*group + 8*(l >> 4) + i][tap] (fused_net.pack_conv_weight_split16); head weights as in bo_tower_s.h (the two 1x1 head
//     convolutions keep their 32x32x16 tiles: 0.5 % of the products);
//   * LDS image [hi | lo][padded 10x10 cell][128] with the 16-byte chunks of a cell at position j ^ 2*(column & 7): a ds_read_b128 is
//     served in four groups of 16 lanes, here 8 positions of one channel octet + 8 positions of the next octet (l >> 4 differs by one);
//     positions of a group span 8 different columns twice, so an EVEN swizzle that is distinct over 8 consecutive columns gives the
//     first 8 lanes the even chunk positions and the other 8 the odd ones: conflict-free for every tap (tests/test_abi.py).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include <type_traits>
#include "bo_tower_s.h"

typedef float bo_f32x4v __attribute__((ext_vector_type(4)));

__device__ __host__ inline int bo_sw16(int cell) { return 2 * ((cell % 10) & 7); }
// offset (in halves) of 16-byte chunk `chunk` (0..15) of padded cell `cell` in a [cell][128] image
__device__ inline int bo_sw16_addr(int cell, int chunk) { return cell * 128 + (((chunk ^ bo_sw16(cell)) & 15) << 3); }

template <int AR = 6>
__global__ void __launch_bounds__(256)
bo_k_tower_s16(const float *__restrict__ x, const bo_h8 *__restrict__ wts, const float *__restrict__ params,
               const bo_tower_layer *__restrict__ layers, int n_layers, float *__restrict__ y, int B, bo_tower_head_s head) {
    constexpr int C = 128, NW = 4, NT = 256, PH = C, CELLS = 100, IMGH = CELLS * PH, CIN0 = 120, HPW = 16 / NW;
    constexpr int UNR = 12;  // K-steps per unrolled body: the three taps of one kernel row x four channel groups
    static_assert(UNR % AR == 0, "the weight ring is indexed statically");
    __shared__ __attribute__((aligned(16))) _Float16 X[2 * IMGH];  // [hi | lo][cell][PH]
    __shared__ __attribute__((aligned(16))) float pooled[C];
    __shared__ float hid[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kb = lane >> 4;   // 16x16x32 operands: column (position) / row (channel) n16, K octet kb
    const int n32 = lane & 31, kg = lane >> 5;   // 32x32x16 operands (head convolutions)
    const int cell0 = ((n16 >> 3) + 1) * 10 + (n16 & 7) + 1;  // padded cell of position n16; position n16 + 16*pt is cell0 + 20*pt
    const int cell32 = ((n32 >> 3) + 1) * 10 + (n32 & 7) + 1;  // of position n32; position n32 + 32 is cell32 + 40

    unsigned long long tseq = 0;  // launch timing: see bo_tower_s.h
    if (head.timing) {
        tseq = __hip_atomic_load(head.timing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0 && tid == 0) head.timing[2 + tseq % BO_TOWER_TIMING_CAP] = wall_clock64();
    }
    for (int i = tid; i < 2 * IMGH / 8; i += NT) reinterpret_cast<bo_h8 *>(X)[i] = bo_h8{0, 0, 0, 0, 0, 0, 0, 0};

    bo_f32x4v acc[2][4];       // [channel tile][position tile]: rows = channels 32*wave + 16*ot + 4*kb + r, col = position 16*pt + n16
    bo_h8 a[AR][2][2];         // weight fragments (channel tile, hi | lo) of AR consecutive K-steps
    bo_h8 bq[2][4][2];         // B operands (position tile, hi | lo) of two consecutive K-steps
    float skip[2][4][4];       // block input at this lane's (channels, positions), float32
    // per column shift dx = -1, 0, 1 of a tap: this lane's chunk index before the channel group is mixed in
    int xk[3];
#pragma unroll
    for (int d = 0; d < 3; d++) xk[d] = kb ^ (2 * (((n16 & 7) + 1 + d - 1) & 7));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bo_h8 *>(wts), 0, 0x7fffffff, 0x00020000);
    const int wvoff = ((wave * 2 * 2) * 64 + lane) * 16;
    auto load_a = [&](int j, int w_off8, int step) {
        typedef int bo_i32x4_t __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane((w_off8 + step * (C / 16) * 2 * 64) * 16);
#pragma unroll
        for (int ot = 0; ot < 2; ot++)
#pragma unroll
            for (int hl = 0; hl < 2; hl++) {
                const bo_i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + (ot * 2 + hl) * 64 * 16, soff, 0);
                a[j][ot][hl] = __builtin_bit_cast(bo_h8, v);
            }
    };
    auto read_b = [&](bo_h8(&b)[4][2], int addr) {  // addr: halves, position tile 0, hi image
        const _Float16 *p = X + addr;
#pragma unroll
        for (int pt = 0; pt < 4; pt++) {
            b[pt][0] = *reinterpret_cast<const bo_h8 *>(p + pt * 20 * PH);
            b[pt][1] = *reinterpret_cast<const bo_h8 *>(p + pt * 20 * PH + IMGH);
        }
    };
#define BO_S16_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

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
                _Float16 *dst = X + bo_sw16_addr(cell + e, ic >> 3) + (ic & 7);
                dst[0] = hi[e]; dst[IMGH] = lo[e];
            }
        }
        __syncthreads();
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = layers[l];
            const bo_tower_layer Ln = layers[l + 1 < n_layers ? l + 1 : 0];
            const int T32 = L.t4 >> 1;  // K-steps of 32 channels (L.t4 counts steps of 16: 72 -> 36)
            const float wscale = params[L.bias_off + C];  // 2^-T: the layer's weights were stored multiplied by 2^T
            float bv[2][4];
#pragma unroll
            for (int ot = 0; ot < 2; ot++)
#pragma unroll
                for (int r = 0; r < 4; r++) bv[ot][r] = params[L.bias_off + wave * 32 + ot * 16 + kb * 4 + r];
#pragma unroll
            for (int ot = 0; ot < 2; ot++)
#pragma unroll
                for (int pt = 0; pt < 4; pt++) acc[ot][pt] = bo_f32x4v{0, 0, 0, 0};
            // step s = 12*row + 4*(dx + 1) + cg: tap (dy = row - 1, dx), channels 32*cg ..
            auto b_addr = [&](int row, int j) {  // K-step j (0..11) of kernel row `row`
                const int dxi = j >> 2, cg = j & 3;
                return (cell0 + (row - 1) * 10 + dxi - 1) * PH + ((xk[dxi] ^ (4 * cg)) << 3);
            };
            read_b(bq[0], b_addr(0, 0));
            for (int row = 0; row < 3; row++) {
#pragma unroll
                for (int j = 0; j < UNR; j++) {
                    const bo_h8(&bc)[4][2] = bq[j & 1];
                    // the step after this one (a harmless re-read at the layer's end)
                    read_b(bq[(j + 1) & 1], j + 1 < UNR ? b_addr(row, j + 1) : b_addr(row < 2 ? row + 1 : 2, row < 2 ? 0 : UNR - 1));
#pragma unroll
                    for (int ot = 0; ot < 2; ot++)
#pragma unroll
                        for (int pt = 0; pt < 4; pt++) {  // the small terms first
                            acc[ot][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j % AR][ot][1], bc[pt][0], acc[ot][pt], 0, 0, 0);
                            acc[ot][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j % AR][ot][0], bc[pt][1], acc[ot][pt], 0, 0, 0);
                            acc[ot][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j % AR][ot][0], bc[pt][0], acc[ot][pt], 0, 0, 0);
                        }
                    const int sn = row * UNR + j + AR;  // this set's next owner: AR steps ahead, maybe in the next layer
                    load_a(j % AR, sn < T32 ? L.w_off4 : Ln.w_off4, sn < T32 ? sn : sn - T32);
                    // every LDS read and weight load in the shadow of different MFMAs
#pragma unroll
                    for (int t = 0; t < 8; t++) { BO_S16_SGB(0x008, 2); BO_S16_SGB(0x100, 1); }
#pragma unroll
                    for (int t = 0; t < 4; t++) { BO_S16_SGB(0x008, 2); BO_S16_SGB(0x020, 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();  // every wave has read the layer input: the output may overwrite it

            // ---- epilogue: a lane holds, per (channel tile, position tile), 4 consecutive channels of one position ----
            float gate[2][4];
            if (L.kind == 3) {
                // SE gate (network.py:33-45).  Wave w owns hidden units w, w + 4, ...; lane n32 owns the gate of channel 32*wave + n32.
                const float *w1 = params + L.se_w1_off, *w2 = params + L.se_w2_off;
                const bool have4 = lane < C / 4;  // a W1 row is C/4 float4: one per lane of the first half
                float w2r[16];
                bo_f32x4 w1r[HPW];
#pragma unroll
                for (int u = 0; u < HPW; u++)
                    w1r[u] = (have4 && wave + u * NW < L.hidden) ? reinterpret_cast<const bo_f32x4 *>(w1 + (size_t)(wave + u * NW) * C)[lane] : bo_f32x4{0, 0, 0, 0};
#pragma unroll
                for (int h = 0; h < 16; h++) w2r[h] = h < L.hidden ? w2[(wave * 32 + n32) * L.hidden + h] : 0.0f;
#pragma unroll
                for (int ot = 0; ot < 2; ot++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float s = bo_row_sum((acc[ot][0][r] + acc[ot][1][r]) + (acc[ot][2][r] + acc[ot][3][r]));  // over the row's 16 positions x 4 tiles
                        if (n16 == 0) pooled[wave * 32 + ot * 16 + kb * 4 + r] = s * (wscale * (1.0f / 64.0f)) + bv[ot][r];
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
                float g = 0.0f;
#pragma unroll
                for (int h = 0; h < 16; h++)
                    if (h < L.hidden) g += w2r[h] * hid[h];
                g = 1.0f / (1.0f + expf(-g));
#pragma unroll
                for (int ot = 0; ot < 2; ot++)
#pragma unroll
                    for (int r = 0; r < 4; r++) gate[ot][r] = __shfl(g, ot * 16 + kb * 4 + r);
            }
            auto write_back = [&](auto kind_c, auto y_c) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr bool TO_Y = decltype(y_c)::value;
                bool sat = false;
#pragma unroll
                for (int ot = 0; ot < 2; ot++)
#pragma unroll
                    for (int pt = 0; pt < 4; pt++) {
                        const int ch0 = wave * 32 + ot * 16 + kb * 4;
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float v = __builtin_fmaf(acc[ot][pt][r], wscale, bv[ot][r]);  // (wscale is a power of two: the product is exact either way)
                            if (KIND == 3) v = v * gate[ot][r];
                            if (KIND >= 2) v += skip[ot][pt][r];
                            o[r] = fmaxf(v, 0.0f);
                            if (KIND != 1) skip[ot][pt][r] = o[r];
                        }
                        bo_h4 hi, lo;
                        bo_split4_pos(o, hi, lo, sat);
                        _Float16 *cellp = X + bo_sw16_addr(cell0 + 20 * pt, ch0 >> 3) + (ch0 & 7);
                        *reinterpret_cast<bo_h4 *>(cellp) = hi;
                        *reinterpret_cast<bo_h4 *>(cellp + IMGH) = lo;
                        if (TO_Y) {
                            float *g2 = y + ((size_t)b * C + ch0) * 64 + pt * 16 + n16;
#pragma unroll
                            for (int r = 0; r < 4; r++) g2[r * 64] = o[r];
                        }
                    }
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
        // ---- the two 1x1 head convolutions + ReLU on the tower output in X (32x32x16 tiles, as in bo_tower_s.h) ----
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
                    const _Float16 *xb = X + bo_sw16_addr(cell32 + 40 * t, 2 * st + kg);
                    const bo_h8 bh = *reinterpret_cast<const bo_h8 *>(xb), bl = *reinterpret_cast<const bo_h8 *>(xb + IMGH);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hacc, 0, 0, 0);
                }
                const int sq = 32 * t + n32;
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
