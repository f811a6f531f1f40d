// betaone_amd/csrc/bo_tower_wg.h -- the residual tower as one persistent kernel, Winograd F(2x2, 3x3) on the fp32
// matrix cores (gfx950).  Same contract as bo_tower.h (one workgroup keeps one board in LDS for the whole tower of
// /root/reference/network.py:48-118,167-190); the 3x3 convolutions cost 16 multiplies per 2x2 output tile instead
// of 36 (2.25x fewer MFMA cycles than the direct form, which runs at 88 % of the fp32 MFMA ceiling).
//
//   Y = A^T [ sum_ic (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// An 8x8 board is 16 tiles, so for each of the 16 Winograd positions the channel sum is a GEMM
// [C_out x C_in] x [C_in x 16 tiles]: v_mfma_f32_16x16x4_f32 with M = 16 output channels, N = the 16 tiles, K = 4
// input channels.
//   * Input transform: fp32 VALU work does not overlap the fp32 MFMAs of the same SIMD (measured: 44 VALU ops per
//     K-step per wave cost exactly their issue time), so it is done ONCE per (channel, tile) and shared: a lane reads
//     the 4x4 patch of its (channel, tile) from the zero-padded 10x10 LDS image, applies B^T d B (32 adds) and
//     stores the 16 positions to an LDS staging buffer V[channel][position quad][tile] that all waves read as their
//     B operand (4 ds_read_b128 per K-step, conflict-free).  V is triple-buffered in chunks of 16 channels
//     (4 K-steps); the chunk two after the current one is transformed by four of the waves while all waves multiply
//     (one workgroup barrier per chunk, in its middle; see `chunk` below).
//   * D lane (n = lane&15, rows 4*(lane>>4)+r) holds, over the 16 position accumulators, the complete 4x4 M matrix of
//     tile n for 4 output channels: A^T M A (24 adds), bias, ReLU, the skip connection (read back from LDS: a lane owns
//     the same (channels, tile) in every layer) and the SE gate happen in registers; the 2x2 result goes straight
//     into the other LDS image.
//   * A operands: the host stores U = G g G^T as Up[step][oc/16][pos/4][lane = 16*(ic&3) + (oc&15)][pos&3], one
//     coalesced global_load_dwordx4 per 4 positions, loaded three K-steps ahead into a rotation of four register sets
//     (96 KB per CU in flight: the weights, 1 MB per layer per CU, stream from L2 at the rate the MFMAs consume them).
// A wave owns 16 output channels (16 position accumulators of 4 registers); C/16 waves per workgroup = two per SIMD at
// C = 128, each under 256 registers.  A K-step is 16 MFMAs of 32 cycles per wave.  MFMA time at 100 % issue, C = 128:
// 32 steps x 2 waves x 16 x 32 cycles = 32.8k cycles = 13.7 us per layer (direct form: 30.7 us).
// fp32 throughout; the transforms use only +/-/x0.5, results agree with the direct form to ~1e-6 relative.
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include <type_traits>
#include "bo_tower.h"

// LAB (scripts/conv_lab.hip only): 0 = the kernel; 1 = no weight loads; 2 = no input transform; 4 = no epilogue (wrong
// results); 5 = no chunk barriers (wrong results); 6 = 1 + 2 + 4 + 5
template <int C, int LAB = 0>
__global__ void __launch_bounds__(C * 4)
bo_k_tower_wg(const float *__restrict__ x, const bo_f32x4 *__restrict__ wts, const float *__restrict__ params,
              const bo_tower_layer *__restrict__ layers, int n_layers, float *__restrict__ y, int B, bo_tower_head head = bo_tower_head{}) {
    constexpr int IMG = 100, NT = C * 4, CP = 128, CIN0 = 120, OB = C / 16;
    // C = 128: the K order of every layer is permuted so that a wave transforms only input channels that it produced itself
    // as the previous layer's output (K-step 4c + sl = channels 16*(4*(c&1) + sl) + 4*(c>>1) + [0,4), the host packs the weights
    // in that order): chunk 0 of the NEXT layer is transformed right behind a wave's own epilogue, before the layer's one
    // workgroup barrier, instead of behind it and in front of a second barrier.
    constexpr bool OWN = C == 128;
    __shared__ __attribute__((aligned(16))) float P[CP * IMG];  // staged input planes / mid activation of a block
    __shared__ __attribute__((aligned(16))) float Q[C * IMG];   // block input
    __shared__ bo_f32x4 V[3][16 * 4 * 16];                      // transformed patches: [buffer][channel 16][pos quad][tile]
    __shared__ float pooled[C], hid[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    const int ty = n >> 2, tx = n & 3;
    const int patch = 20 * ty + 2 * tx;       // top-left of the tile's 4x4 input patch in a padded 10x10 image
    const int ocell = patch + 11;             // top-left of its 2x2 output block
    const int oc0 = wave * 16 + 4 * kq;       // this lane's 4 output channels: oc0 + r

    // the first board's input planes are requested before anything else: their latency hides behind the zero fill
    constexpr int XQ = 128 * 16 / NT;
    auto fetch_board = [&](bo_f32x4(&xin)[XQ], int b) {
        const bo_f32x4 *xb = reinterpret_cast<const bo_f32x4 *>(x + (size_t)b * CIN0 * 64);
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            const int i = tid + q * NT;
            xin[q] = i < CIN0 * 16 ? xb[i] : bo_f32x4{0, 0, 0, 0};
        }
    };
    auto stage_board = [&](const bo_f32x4(&xin)[XQ]) {  // into P (channels 120..127 of the padded input conv stay zero)
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            const int i = tid + q * NT;
            const int ic = i >> 4, q16 = i & 15, r = q16 >> 1, c0 = (q16 & 1) * 4;
            float *dst = P + ic * IMG + (r + 1) * 10 + c0 + 1;
            dst[0] = xin[q][0]; dst[1] = xin[q][1]; dst[2] = xin[q][2]; dst[3] = xin[q][3];
        }
    };
    {
        bo_f32x4 xin[XQ];
        if ((int)blockIdx.x < B) fetch_board(xin, blockIdx.x);
        for (int i = tid; i < CP * IMG / 4; i += NT) reinterpret_cast<bo_f32x4 *>(P)[i] = bo_f32x4{0, 0, 0, 0};
        for (int i = tid; i < C * IMG / 4; i += NT) reinterpret_cast<bo_f32x4 *>(Q)[i] = bo_f32x4{0, 0, 0, 0};
        __syncthreads();
        if ((int)blockIdx.x < B) stage_board(xin);
    }

    bo_f32x4 acc[16];
    bo_f32x4 bias4 = {0, 0, 0, 0};
    bo_f32x4 a0[4], a1[4], a2[4], a3[4];  // A fragments of four consecutive K-steps: [position quad]
    bo_f32x4 va[4], vb[4];                // B operands of two consecutive K-steps: [position quad]
    // Weight fragments come through a buffer descriptor: the per-thread part of the address (wave, lane) is ONE constant
    // VGPR, the per-step part (layer offset, K-step) is scalar arithmetic, the position quad is the instruction's immediate
    // offset -- no vector ALU work per load (as 64-bit global pointers every K-step cost three VALU address operations,
    // which on this kernel come straight out of the MFMA issue time).
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bo_f32x4 *>(wts), 0, 0x7fffffff, 0x00020000);
    const int wvoff = ((wave * 4) * 64 + lane) * 16;
    auto ldw = [&](int w_off4, int step, int pq) -> bo_f32x4 {
        typedef int bo_i32x4_t __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane((w_off4 + step * (OB * 4 * 64)) * 16);
        const bo_i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + pq * 64 * 16, soff, 0);
        return __builtin_bit_cast(bo_f32x4, v);
    };
    auto load_w = [&](bo_f32x4(&a)[4], int w_off4, int step) {
#pragma unroll
        for (int pq = 0; pq < 4; pq++) a[pq] = ldw(w_off4, step, pq);
    };
    // chunk c of a layer = input channels 16c..16c+15; at C = 128 the even chunks are transformed by waves 0-3 and the
    // odd ones by waves 4-7 (one of each per SIMD), at C = 64 by all four waves
    auto my_chunk = [&](int c) { return C == 128 ? (wave >> 2) == (c & 1) : true; };
    auto transform = [&](const float *img, int c, int vbuf) {  // V[vbuf][channel][.][tile] = B^T d B of chunk c, one (channel, tile) per lane
        const int icl = 4 * (wave & 3) + kq;  // slot in the chunk: K-step wave & 3, row kq
        const float *p = img + (OWN ? 16 * wave + 4 * (c >> 1) + kq : 16 * c + icl) * IMG + patch;
        // 32 additions as 16 packed ones (v_pk_add_f32 with per-lane operand selection and negation): vector ALU work does
        // not overlap this SIMD's MFMAs, every instruction here comes out of the other wave's matrix time.
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 dl[4], dh[4], wl[4], wh[4];  // (x0, x1) and (x2, x3) of each patch row
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dl[i] = *reinterpret_cast<const f2 *>(p + 10 * i);
            dh[i] = *reinterpret_cast<const f2 *>(p + 10 * i + 2);
        }
        wl[0] = dl[0] - dl[2]; wh[0] = dh[0] - dh[2];   // B^T d: rows (0-2, 1+2, 2-1, 1-3), both column pairs at once
        wl[1] = dl[1] + dl[2]; wh[1] = dh[1] + dh[2];
        wl[2] = dl[2] - dl[1]; wh[2] = dh[2] - dh[1];
        wl[3] = dl[1] - dl[3]; wh[3] = dh[1] - dh[3];
        f2 *dst = reinterpret_cast<f2 *>(&V[vbuf][icl * 64 + n]);
#pragma unroll
        for (int i = 0; i < 4; i++) {  // (.) B per row: (w0 - w2, w1 + w2) and (w2 - w1, w1 - w3), one instruction each
            f2 v01, v23;
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(wl[i]), "v"(wh[i]));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(v23) : "v"(wl[i]), "v"(wh[i]));
            dst[i * 32] = v01;      // V[.][channel][position quad i][tile]: 16 bytes per (quad, tile), 16 tiles per quad
            dst[i * 32 + 1] = v23;
        }
    };
    // max(x, 0) as the one instruction it is (fmaxf() first canonicalises its operand: a second v_max_f32 per value)
    auto relu = [](float x) {
        float y;
        asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
        return y;
    };
    auto read_b = [&](bo_f32x4(&v)[4], int buf, int sl) {  // B operands of local step sl (channels 4*sl + kq of the chunk)
        const bo_f32x4 *src = &V[buf][(4 * sl + kq) * 64 + n];
#pragma unroll
        for (int pq = 0; pq < 4; pq++) v[pq] = src[pq * 16];
    };
    // One K-step = 16 MFMAs in four blocks of 4 (position quad pq).  Each block also issues the weight load of the same
    // quad two steps ahead and the B operand read of the same quad of the next step; the sched_group_barriers put each
    // into the shadow of a different MFMA (the scheduler's own order issues loads right before their use).
#define BO_WG_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    // first = the layer's first K-step: its MFMAs take a zero C operand (an inline constant) instead of accumulators that 64
    // v_mov instructions per wave and layer would have had to clear
    // (vbuf_next, sl_next): where the B operands of the NEXT K-step are (the last K-step of a chunk reads the first of the next chunk)
    auto kstep = [&](const bo_f32x4(&a)[4], const bo_f32x4(&v)[4], bo_f32x4(&an)[4], bo_f32x4(&vn)[4], int vbuf_next, int sl_next, int w_off4n,
                     int step_w, auto first) {
        const bo_f32x4 *src = &V[vbuf_next][(4 * sl_next + kq) * 64 + n];
#pragma unroll
        for (int pq = 0; pq < 4; pq++) {
            if (LAB != 1 && LAB != 6) an[pq] = ldw(w_off4n, step_w, pq);
            vn[pq] = src[pq * 16];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if constexpr (decltype(first)::value)  // position 5 = M[1][1] enters every output of A^T M A once: it starts at the bias
                    acc[4 * pq + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[pq][e], v[pq][e], 4 * pq + e == 5 ? bias4 : bo_f32x4{0, 0, 0, 0}, 0, 0, 0);
                else
                    acc[4 * pq + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[pq][e], v[pq][e], acc[4 * pq + e], 0, 0, 0);
            }
            BO_WG_SGB(0x008, 1); BO_WG_SGB(0x020, 1); BO_WG_SGB(0x008, 1); BO_WG_SGB(0x100, 1); BO_WG_SGB(0x008, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- the 120 input planes of board b are in P (the first board's: staged above) ----
        if (b != (int)blockIdx.x) {
            bo_f32x4 xin[XQ];
            fetch_board(xin, b);
            __syncthreads();  // the previous board's readers of P are done
            stage_board(xin);
        } else {
            load_w(a0, layers[0].w_off4, 0);  // (a later board finds them loaded by the previous board's last K-steps)
            load_w(a1, layers[0].w_off4, 1);
            load_w(a2, layers[0].w_off4, 2);
        }
        __syncthreads();
        // chunks 0 and 1 of the input convolution (every later layer's: behind the previous layer's epilogue)
        int vb0 = 0;  // V buffer of the current layer's chunk 0; chunk c is in buffer (vb0 + c) % 3
        if (LAB != 2 && LAB != 6) {
            if (my_chunk(0)) transform(P, 0, 0);
            if (my_chunk(1)) transform(P, 1, 1);
        }
        __syncthreads();
        bo_tower_layer Lnext = layers[0];
        bias4 = *reinterpret_cast<const bo_f32x4 *>(params + Lnext.bias_off + oc0);  // this lane's 4 output channels, first layer
        for (int l = 0; l < n_layers; l++) {
            const bo_tower_layer L = Lnext;  // (fetched one layer ahead: no scalar-load latency at the top of a layer)
            const bo_tower_layer Ln = Lnext = layers[l + 1 < n_layers ? l + 1 : 0];
            const float *img = L.kind == 1 ? Q : P;
            // One chunk = 4 K-steps = one turn of the weight-set rotation: step s multiplies with set s%4 while the weights
            // of step s+3 (possibly the first steps of the next layer) are loaded into set (s+3)%4.
            const int nk = L.t4, nchunks = nk >> 2;
            // V is triple-buffered and a chunk's transform runs TWO chunks ahead, behind the one workgroup barrier in the middle
            // of a chunk: at that barrier chunk c+1's operands (transformed during chunk c-1) are complete, and every wave is
            // done with chunk c-1's buffer, which chunk c+2's transform overwrites.  The B operands of a chunk's first K-step are
            // read by the previous chunk's last K-step: no barrier is followed by an exposed LDS read inside a layer.
            auto chunk = [&](int c, int vbuf, auto first) {
                const int s = 4 * c, t4 = s + 4, t5 = s + 5, t6 = s + 6;
                const int vbuf1 = vbuf == 2 ? 0 : vbuf + 1, vbuf2 = vbuf1 == 2 ? 0 : vbuf1 + 1;
                kstep(a0, va, a3, vb, vbuf, 1, L.w_off4, s + 3, first);
                kstep(a1, vb, a0, va, vbuf, 2, t4 < nk ? L.w_off4 : Ln.w_off4, t4 < nk ? t4 : t4 - nk, std::false_type{});
                if (LAB != 5 && LAB != 6) __syncthreads();
                if (LAB != 2 && LAB != 6 && c + 2 < nchunks && my_chunk(c + 2)) {
                    __builtin_amdgcn_s_setprio(3);  // its vector instructions back to back, not alternating with the other wave's MFMAs
                    transform(img, c + 2, vbuf2);
                    __builtin_amdgcn_s_setprio(0);
                }
                kstep(a2, va, a1, vb, vbuf, 3, t5 < nk ? L.w_off4 : Ln.w_off4, t5 < nk ? t5 : t5 - nk, std::false_type{});
                kstep(a3, vb, a2, va, vbuf1, 0, t6 < nk ? L.w_off4 : Ln.w_off4, t6 < nk ? t6 : t6 - nk, std::false_type{});
            };
            read_b(va, vb0, 0);
            __builtin_amdgcn_sched_barrier(0);
            int vbc = vb0;
            chunk(0, vbc, std::true_type{});
            bias4 = *reinterpret_cast<const bo_f32x4 *>(params + Ln.bias_off + oc0);  // consumed by the first K-step only: the next layer's, now
            for (int c = 1; c < nchunks; c++) {
                vbc = vbc == 2 ? 0 : vbc + 1;
                chunk(c, vbc, std::false_type{});
            }
            vb0 = vbc == 2 ? 0 : vbc + 1;  // the next layer's chunk 0

            // ---- output transform Y = A^T M A per row r: M[i][j] = acc[4i+j][r] ----
            // (two output channels per instruction: the accumulator registers of rows r, r+1 are adjacent -> v_pk_add_f32;
            //  the same operations in the same order per element as the scalar form)
            float o[4][4];
#pragma unroll
            for (int p2 = 0; p2 < 2; p2++) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                // x - y on both halves as ONE v_pk_add_f32 with negated second operand (the compiler emits two v_sub_f32)
                auto psub = [](f2 x, f2 y) {
                    f2 d;
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
                    return d;
                };
                f2 t0[4], t1[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f2 a0 = {acc[j][2 * p2], acc[j][2 * p2 + 1]}, a1 = {acc[4 + j][2 * p2], acc[4 + j][2 * p2 + 1]};
                    const f2 a2 = {acc[8 + j][2 * p2], acc[8 + j][2 * p2 + 1]}, a3 = {acc[12 + j][2 * p2], acc[12 + j][2 * p2 + 1]};
                    t0[j] = a0 + a1 + a2;
                    t1[j] = psub(psub(a1, a2), a3);
                }
                const f2 o0 = t0[0] + t0[1] + t0[2], o1 = psub(psub(t0[1], t0[2]), t0[3]);  // (the bias came in with acc[5])
                const f2 o2 = t1[0] + t1[1] + t1[2], o3 = psub(psub(t1[1], t1[2]), t1[3]);
                o[2 * p2][0] = o0[0]; o[2 * p2 + 1][0] = o0[1];
                o[2 * p2][1] = o1[0]; o[2 * p2 + 1][1] = o1[1];
                o[2 * p2][2] = o2[0]; o[2 * p2 + 1][2] = o2[1];
                o[2 * p2][3] = o3[0]; o[2 * p2 + 1][3] = o3[1];
            }
            if (LAB == 4 || LAB == 6) {
                if (o[0][0] == 123.456f) Q[tid] = o[1][1] + o[2][2] + o[3][3];
            } else if (L.kind <= 1) {
                float *out = L.kind == 0 ? Q : P;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float *c = out + (oc0 + r) * IMG + ocell;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[r][e] = relu(o[r][e]);
                    c[0] = o[r][0]; c[1] = o[r][1]; c[10] = o[r][2]; c[11] = o[r][3];
                }
            } else {
                float gate[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (L.kind == 3) {
                    // SE gate (network.py:33-45).  Weights first, so their latency hides behind the pooling: wave h owns hidden
                    // unit h (W1 row h at channels lane, lane+64), lane (lane&15) of wave w owns the gate of channel 16w + (lane&15)
                    const float *w1 = params + L.se_w1_off, *w2 = params + L.se_w2_off;
                    constexpr int NWAVE = NT / 64;
                    float w1a[2] = {0.0f, 0.0f}, w1b[2] = {0.0f, 0.0f}, w2r[16];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int h = wave + u * NWAVE;
                        if (h < L.hidden) { w1a[u] = w1[h * C + lane]; w1b[u] = C > 64 ? w1[h * C + 64 + lane] : 0.0f; }
                    }
#pragma unroll
                    for (int h = 0; h < 16; h++) w2r[h] = h < L.hidden ? w2[(wave * 16 + n) * L.hidden + h] : 0.0f;
                    // channel means of conv + bias over the 64 squares = 16 tiles x 4 outputs (AdaptiveAvgPool2d(1))
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float s = bo_row_sum((o[r][0] + o[r][1]) + (o[r][2] + o[r][3]));  // over the 16 tiles (DPP)
                        if (n == 0) pooled[oc0 + r] = s * (1.0f / 64.0f);
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < 2; u++) {  // hidden = relu(W1 mean): one wave reduction per hidden unit
                        const int h = wave + u * NWAVE;
                        if (h < L.hidden) {
                            const float p = bo_wave_sum63(w1a[u] * pooled[lane] + (C > 64 ? w1b[u] * pooled[64 + lane] : 0.0f));
                            if (lane == 63) hid[h] = fmaxf(p, 0.0f);
                        }
                    }
                    __syncthreads();
                    float a = 0.0f;
#pragma unroll
                    for (int h = 0; h < 16; h++)
                        if (h < L.hidden) a += w2r[h] * hid[h];
                    const float g = 1.0f / (1.0f + expf(-a));
#pragma unroll
                    for (int r = 0; r < 4; r++) gate[r] = __shfl(g, 4 * kq + r);
                }
                float *yb = y + (size_t)b * C * 64;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int oc = oc0 + r;
                    float *c = Q + oc * IMG + ocell;
                    // the skip connection: the block's input at this lane's (channel, tile) is what this lane stored here two
                    // layers ago (kept in registers it cost 16 VGPRs through both convolutions' K loops)
                    const float skip[4] = {c[0], c[1], c[10], c[11]};
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) t[e] = relu(L.kind == 3 ? o[r][e] * gate[r] + skip[e] : o[r][e] + skip[e]);
                    c[0] = t[0]; c[1] = t[1]; c[10] = t[2]; c[11] = t[3];
                    if (L.last && y) {
                        float *g2 = yb + oc * 64 + 16 * ty + 2 * tx;
                        *reinterpret_cast<float2 *>(g2) = float2{t[0], t[1]};
                        *reinterpret_cast<float2 *>(g2 + 8) = float2{t[2], t[3]};
                    }
                }
            }
            if (l + 1 < n_layers && LAB != 2 && LAB != 6) {
                float *nimg = L.kind == 1 ? P : Q;  // the image just written = the next layer's input
                const int vb1 = vb0 == 2 ? 0 : vb0 + 1;
                if (OWN) {
                    // this wave's output channels are the next layer's input channels of this wave's chunks: LDS operations of one
                    // wave execute in order, so its reads below see its writes above without a barrier.  (Buffers: the last chunk
                    // is still being read by slower waves from (vb0 + 2) % 3; vb0 and vb1 were last read before its barrier.)
                    __builtin_amdgcn_wave_barrier();
                    transform(nimg, my_chunk(0) ? 0 : 1, my_chunk(0) ? vb0 : vb1);
                } else {
                    __syncthreads();
                    transform(nimg, 0, vb0);
                    transform(nimg, 1, vb1);
                }
            }
            __syncthreads();
        }
        // ---- the two 1x1 head convolutions + ReLU (network.py:101-113,191-195) on the tower output still in Q:
        // D[oc][square] = sum_ic Wh[oc][ic] * Q[ic][square], one 16x16 job per (16 head channels, 16 squares) ----
        if (head.channels > 0) {
            // head weights, packed on the host: Whp[mb][g][lane][e] = Wh[16*mb + (lane&15)][4*(4*g + e) + (lane>>4)] (0 beyond the
            // last channel): the A fragments of four K-steps per global_load_dwordx4, all issued before the first MFMA
            const bo_f32x4 *whp = reinterpret_cast<const bo_f32x4 *>(params + head.w_off);
            const float *bh = params + head.b_off;
            const int jobs = ((head.channels + 15) >> 4) * 4;
            for (int job = wave; job < jobs; job += NT / 64) {
                const int mb = job >> 2, nb = job & 3, sq = 16 * nb + n, cell = ((sq >> 3) + 1) * 10 + (sq & 7) + 1;
                const int oc_d = 16 * mb + 4 * kq;
                bo_f32x4 aw[C / 16];
#pragma unroll
                for (int g = 0; g < C / 16; g++) aw[g] = whp[(mb * (C / 16) + g) * 64 + lane];
                bo_f32x4 hacc = {0, 0, 0, 0};
#pragma unroll
                for (int g = 0; g < C / 16; g++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g][e], Q[(4 * (4 * g + e) + kq) * IMG + cell], hacc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int oc = oc_d + r;
                    if (oc < head.channels) {
                        const float t = fmaxf(hacc[r] + bh[oc], 0.0f);
                        if (oc < head.split) head.out_a[((size_t)b * head.split + oc) * 64 + sq] = t;
                        else head.out_b[((size_t)b * (head.channels - head.split) + (oc - head.split)) * 64 + sq] = t;
                    }
                }
            }
        }
    }
}
#endif
