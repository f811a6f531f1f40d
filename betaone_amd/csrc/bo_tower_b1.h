// betaone_amd/csrc/bo_tower_b1.h -- the residual tower of ONE board (up to a few) as ONE launch spread over the chip (gfx950).
//
// BASELINE.json configs[3]: uci.py analyses a single position (/root/reference/uci.py:60-93 -> run_mcts, mcts.py:155-280), so every
// evaluation of the search is PolicyValueNet.forward (/root/reference/network.py:167-198) at batch 1, 18 of them in sequence per
// 1600-simulation search.  As one launch per layer (bo_k_conv3x3_small, bo_conv.h: a board's layer cut into (C/16) x 4 workgroups)
// the 41 layers of the 15+5 x 256 net are 41 dependent launches of 8.6 us + 5 SE launches: 0.47 ms per evaluation, launch-bound.
// Here the whole tower is one launch of the same (C/16) x 4 workgroups per board; the kernel boundary between two layers becomes a
// hand-off inside the launch:
//   * a workgroup owns the output tile (16 channels x 16 squares = two board rows) of EVERY layer; its four waves split K (a quarter
//     of the input channels each, all 9 taps, v_mfma_f32_16x16x4_f32: exact float32 products as in bo_k_conv3x3_small, the same
//     summation order, so layers without an SE gate are bit-identical to that route) and reduce through LDS;
//   * the tile is stored WRITE-THROUGH (16-byte `sc1` stores: a lane holds four consecutive squares of one channel -- the MFMA
//     operands are swapped against bo_k_conv3x3_small for that), the storing wave drains them (s_waitcnt vmcnt(0)) and one lane adds
//     to the board's arrival counter (agent scope); a consumer polls that ONE word with `sc1` loads, runs ONE agent-scope acquire
//     (it drops this CU's L1 lines: L1 is never refreshed by other CUs' stores), waits for it, joins the workgroup barrier, then every
//     wave reads its slab with plain 16-byte loads (cdna_hip_programming.md Guideline 16: R1 producer, "Consumer, always" form).
//     Measured alternatives (BO_B1_SC1 / BO_B1_NT, profiles/r04_b1_tower.md): `sc1` loads of every handed-off byte and no acquire --
//     valid (MI355X_MICROARCH.md, visibility, table row 1) but 87 us SLOWER per evaluation: the workgroups of an XCD that read the same
//     input rows no longer share them through its L2; the acquire issued EARLY (right after the layer's own slab loads, so that it
//     would run under the MFMA phase): 100 us slower -- the invalidate sits in front of the tile's store in the CU's memory pipeline and
//     delays the arrival by its own length.  HIP promises nothing about placement: nothing here depends on which XCD a workgroup runs on;
//   * the next layer's weight fragments (9 taps x K/64 float4 per lane, all of them in registers) are requested BEFORE the wait, so
//     the weight stream's latency (100 MB per evaluation from the Infinity Cache / HBM) hides under the hand-off;
//   * SE blocks (network.py:33-45) exchange 4 x C partial channel sums instead of the layer: the un-gated tile stays in the
//     registers of the wave that made it, every workgroup computes the gate of its own 16 channels after the hand-off;
//   * every spin is bounded: a workgroup that waits longer than BO_B1_SPIN_LIMIT polls writes a code into the status word and
//     every wave of the grid leaves the kernel (bo_nn_b1_status reports it) -- no wave can wait for ever.
// The grid must be resident at once: (C/16) * 4 * batch <= 256 workgroups of 256 threads, one per CU (checked by bo_nn_b1_forward).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_conv.h"

#define BO_B1_SPIN_LIMIT (1u << 18)   // polls of ~1 us each before a wait gives up (a hand-off takes a few us)
#define BO_B1_MAX_LAYERS 96
// How a consumer reads bytes other workgroups stored in this launch (bo_nn_b1_forward: BO_B1_SC1 unless the environment variable
// BETAONE_B1_MODE names another; measured per evaluation of the 15+5 x 256 tower, fp32 MFMA: ACQ 308 us, SC1 288 us, NT 320 us):
enum { BO_B1_ACQ = 0,   // poll -> ONE agent-scope acquire (buffer_inv sc1: this CU's L1 dropped) -> its wait -> barrier -> plain loads
       BO_B1_SC1 = 1,   // poll -> barrier -> every load `sc1` (past L1), no acquire
       BO_B1_NT = 2 };  // poll -> barrier -> every load `nt` (past L1, L2-served), no acquire

struct bo_b1_layer {        // one 3x3 convolution of the tower (device table)
    const bo_f32x4 *w;      // fused_net.pack_conv_weight_small: [C/16][tap 9][cin/16][64][4]
    const float *bias;      // [C]
    const float *se_w1;     // [H][C]  (mode 2)
    const float *se_w2;     // [C][H]
    int cin, cin_x;         // K channels the weights are packed for (128 for the input layer); channels present in the input
    int mode;               // 0: relu(conv + bias)   1: relu(conv + bias + residual)   2: relu((conv + bias) * se_gate + residual)
    int se_h;
    const bo_f32x4 *w_split;  // BO_B1_SPLIT: fused_net.pack_conv_weight_small_split: [C/16][tap 9][cin/16][64][hi x4 | lo x4] fp16 of scale * w
    float inv_scale;        // 1 / scale (a power of two)
    int pad;
};

struct bo_b1_args {
    const float *x;         // [B][120][64] input planes
    float *y;               // [B][C][64] tower output
    float *bufs;            // [3][B][C][64] activations between layers
    float *pool;            // [B][4][C] partial channel sums of an SE layer (one row per position tile)
    unsigned *sync;         // [0..MB): arrival counters of board b, NEVER reset: a launch counts on from the value it finds; [MB..2MB): that
                            // value = the board's counter when its last launch ended (written by the launch's last arriver); [2MB]: status word
                            // (0 ok, else 1 + phase of the wait that gave up); [2MB + 1]: 1 if an activation left the fp16 range (BO_B1_SPLIT).
                            // The host zeroes the block when the handle is made and after a fault -- no memset rides in front of a launch
                            // (three launches captured in ONE graph ran their memset nodes into each other's kernels: profiles/r04_b1_tower.md).
    int MB;                 // the handle's max_batch
    const bo_b1_layer *layers;
    int n_layers, B;
    unsigned long long *prof;   // NULL, or [B * tiles][4 waves][8]: shader-clock sums per wave of {wait, stage, mfma, reduce, epilogue, layers}
};

typedef __attribute__((address_space(1))) unsigned bo_gu32;
#define BO_GU32(p) ((bo_gu32 *)(p))   // a generic pointer known to be global: GLOBAL (not flat) agent-scope accesses

__device__ __forceinline__ void bo_b1_store16_sc1(float *p, bo_f32x4 v) {  // write-through: the bytes leave this XCD's L2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bo_b1_store4_sc1(float *p, float v) {
    __hip_atomic_store(BO_GU32(reinterpret_cast<unsigned *>(p)), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// loads of bytes another workgroup stored in this launch (behind bo_b1_wait): `sc1` buffer loads to registers (compiler-visible, so
// their waits are the compiler's), addressed by byte offset into the activation buffers' descriptor
typedef int bo_b1_i32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__device__ __forceinline__ bo_f32x4 bo_b1_load16(__amdgpu_buffer_rsrc_t rsrc, const float *base, size_t float_off) {
    if constexpr (MODE == BO_B1_ACQ) return *reinterpret_cast<const bo_f32x4 *>(base + float_off);
    else if constexpr (MODE == BO_B1_NT) return __builtin_nontemporal_load(reinterpret_cast<const bo_f32x4 *>(base + float_off));
    else return __builtin_bit_cast(bo_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(float_off * 4), 0, 16));  // aux 16 = sc1
}
template <int MODE>
__device__ __forceinline__ float bo_b1_load4(const float *p) {
    if constexpr (MODE == BO_B1_ACQ) return *p;
    else if constexpr (MODE == BO_B1_NT) return __builtin_nontemporal_load(p);
    else return __uint_as_float(__hip_atomic_load(BO_GU32(reinterpret_cast<const unsigned *>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The storing wave (wave 0) has issued its write-through stores: drain them, then ONE lane signals.
__device__ __forceinline__ void bo_b1_arrive(unsigned *ctr, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(BO_GU32(ctr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The launch's LAST hand-off of a board (nobody waits for it inside the launch): the arriver that completes it notes the counter's
// value for the next launch to count on from.
__device__ __forceinline__ void bo_b1_arrive_last(unsigned *ctr, unsigned *base_word, unsigned final_value, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(BO_GU32(ctr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == final_value) __hip_atomic_store(BO_GU32(base_word), final_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Every wave calls this; ONE lane (tid 64: wave 1, which stores nothing) issues the acquire, polls the board's counter (`sc1` loads,
// bounded), waits for the invalidate, then the workgroup barrier; afterwards loads of the handed-off bytes are valid in every wave.  false: the wait gave up (status word written) or
// another workgroup did -- uniform over the workgroup.
template <int MODE>
__device__ __forceinline__ bool bo_b1_wait(unsigned *ctr, unsigned *status, unsigned target, unsigned code, int tid, int *ok_lds) {
    if (tid == 64) {
        bool ok = true;
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(BO_GU32(ctr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {  // (wrap-safe: counters run on for ever)
            __builtin_amdgcn_s_sleep(2);
            if (++spins > BO_B1_SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(BO_GU32(status), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (spins > BO_B1_SPIN_LIMIT) __hip_atomic_store(BO_GU32(status), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        if constexpr (MODE == BO_B1_ACQ) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the compiler from moving loads above the poll)
        }
        *ok_lds = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// One convolution of one tile: stage this wave's slab, its share of K on the matrix pipe, reduction over the four waves.
// Returns (in wave 0) conv + bias for squares 16*pt + 4*kq + [0,4) of channel 16*ot + n.
template <int CIN, int GM, bool HANDED, int MODE, int C>   // HANDED: the input was stored by other workgroups of this launch (not the kernel's input planes)
__device__ __forceinline__ bo_f32x4 bo_b1_conv(const float *xin, __amdgpu_buffer_rsrc_t rsrc, const float *bufs, int cin_x, const bo_f32x4 (&a)[9][GM], float *Xw,
                                               bo_f32x4 (*red)[64], const float *__restrict__ bias, int ot, int pt, int wave, int lane, unsigned long long *tm) {
    constexpr int CQ = CIN / 4, G = CQ / 16, SLAB = 40, NST = CQ / 8;
    const int kq = lane >> 4, n = lane & 15;
    {   // slab: channels wave*CQ .. +CQ, board rows 2pt-1 .. 2pt+2 (32 contiguous floats per channel where the rows exist), halo columns stay zero
        // Every lane loads (a clamped address where its cell is outside the board), ALL loads are issued before the first is consumed,
        // the select comes with the LDS store.  (A load under a lane condition became a branch per load; a select right behind each
        // load a wait per load: either way the eight loads of a slab were eight dependent round trips, +35..100 us per evaluation.)
        bo_f32x4 st[NST];
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i >> 3, q = i & 7, row = 2 * pt - 1 + (q >> 1);
            const bool in = row >= 0 && row < 8 && wave * CQ + c < cin_x;
            const size_t e = in ? (size_t)(wave * CQ + c) * 64 + row * 8 + (q & 1) * 4 : 0;
            st[u] = HANDED ? bo_b1_load16<MODE>(rsrc, bufs, (size_t)(xin - bufs) + e) : *reinterpret_cast<const bo_f32x4 *>(xin + e);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i >> 3, q = i & 7, row = 2 * pt - 1 + (q >> 1);
            const bool in = row >= 0 && row < 8 && wave * CQ + c < cin_x;
            float *d = Xw + c * SLAB + (q >> 1) * 10 + 1 + (q & 1) * 4;
            d[0] = in ? st[u][0] : 0.0f; d[1] = in ? st[u][1] : 0.0f; d[2] = in ? st[u][2] : 0.0f; d[3] = in ? st[u][3] : 0.0f;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the slab is this wave's own (no workgroup barrier needed)
    __builtin_amdgcn_wave_barrier();
    if (tm) tm[0] = __builtin_amdgcn_s_memtime();
    const float *xl = Xw + kq * SLAB + (n >> 3) * 10 + (n & 7);
    bo_f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int off = (tap / 3) * 10 + tap % 3;
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int e = 0; e < 4; e++)  // M = squares (A = activations), N = channels (B = weights): a lane's four results are consecutive squares
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xl[(4 * g + e) * 4 * SLAB + off], a[tap][g][e], acc, 0, 0, 0);
    }
    if (tm) { asm volatile("s_nop 0" :: "v"(acc)); tm[1] = __builtin_amdgcn_s_memtime(); }
    red[wave][lane] = acc;
    __syncthreads();
    if (tm) tm[2] = __builtin_amdgcn_s_memtime();
    bo_f32x4 v = {0, 0, 0, 0};
    if (wave == 0) {
        const bo_f32x4 s0 = red[0][lane], s1 = red[1][lane], s2 = red[2][lane], s3 = red[3][lane];
        const float bc = bias[16 * ot + n];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = ((s0[r] + s1[r]) + (s2[r] + s3[r])) + bc;
    }
    return v;
}

// The same tile on the fp16 matrix pipe (precision = the split-precision tower's, bo_tower_s.h): every float32 operand a (hi, lo) pair
// of fp16 numbers (hi = RN16(v), lo = RN16(v - hi): 22 significant bits), every product three v_mfma_f32_16x16x16_f16 (lo x hi + hi x lo
// + hi x hi) accumulated in float32: 27 matrix-pipe cycles per 16 input channels instead of the 128 of four v_mfma_f32_16x16x4_f32.
// The slab holds the pairs as two images [channel quad][cell][4 channels] of fp16 (one ds_read_b64 = the A fragment of a K-step); the
// split happens while the slab is staged.  A value beyond the fp16 range cannot be carried: it is saturated and *overflow set (the
// evaluation is then wrong and bo_nn_b1_status says so).
typedef _Float16 bo_b1_h4 __attribute__((ext_vector_type(4)));
template <int CIN, int GM, bool HANDED, int MODE, int C>
__device__ __forceinline__ bo_f32x4 bo_b1_conv_split(const float *xin, __amdgpu_buffer_rsrc_t rsrc, const float *bufs, int cin_x, const bo_f32x4 (&a)[9][GM], float *Xw,
                                                     bo_f32x4 (*red)[64], const float *__restrict__ bias, float inv_scale, int ot, int pt, int wave, int lane, bool &overflow, unsigned long long *tm) {
    constexpr int CQ = CIN / 4, G = CQ / 16, SLAB = 40, NST = CQ / 8;
    const int kq = lane >> 4, n = lane & 15;
    _Float16 *Xh = reinterpret_cast<_Float16 *>(Xw), *Xl = Xh + (CQ / 4) * SLAB * 4;  // [CQ/4][SLAB][4] each: 2 x CQ x SLAB x 2 B = the float32 slab's bytes
    {
        bo_f32x4 st[NST];
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i >> 3, q = i & 7, row = 2 * pt - 1 + (q >> 1);
            const bool in = row >= 0 && row < 8 && wave * CQ + c < cin_x;
            const size_t e = in ? (size_t)(wave * CQ + c) * 64 + row * 8 + (q & 1) * 4 : 0;
            st[u] = HANDED ? bo_b1_load16<MODE>(rsrc, bufs, (size_t)(xin - bufs) + e) : *reinterpret_cast<const bo_f32x4 *>(xin + e);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i >> 3, q = i & 7, row = 2 * pt - 1 + (q >> 1);
            const bool in = row >= 0 && row < 8 && wave * CQ + c < cin_x;
            const int base = ((c >> 2) * SLAB + (q >> 1) * 10 + 1 + (q & 1) * 4) * 4 + (c & 3);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = in ? st[u][j] : 0.0f;
                if (fabsf(v) > 65504.0f) { overflow = true; v = v > 0.0f ? 65504.0f : -65504.0f; }
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                Xh[base + 4 * j] = hi;
                Xl[base + 4 * j] = lo;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (tm) tm[0] = __builtin_amdgcn_s_memtime();
    const bo_b1_h4 *xh = reinterpret_cast<const bo_b1_h4 *>(Xh) + kq * SLAB + (n >> 3) * 10 + (n & 7);
    const bo_b1_h4 *xl = reinterpret_cast<const bo_b1_h4 *>(Xl) + kq * SLAB + (n >> 3) * 10 + (n & 7);
    bo_f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int off = (tap / 3) * 10 + tap % 3;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const bo_b1_h4 ah = xh[4 * g * SLAB + off], al = xl[4 * g * SLAB + off];
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 w8 = __builtin_bit_cast(h8, a[tap][g]);
            const bo_b1_h4 wh = {w8[0], w8[1], w8[2], w8[3]}, wl = {w8[4], w8[5], w8[6], w8[7]};
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al, wh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, wl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, wh, acc, 0, 0, 0);
        }
    }
    if (tm) { asm volatile("s_nop 0" :: "v"(acc)); tm[1] = __builtin_amdgcn_s_memtime(); }
    red[wave][lane] = acc;
    __syncthreads();
    if (tm) tm[2] = __builtin_amdgcn_s_memtime();
    bo_f32x4 v = {0, 0, 0, 0};
    if (wave == 0) {
        const bo_f32x4 s0 = red[0][lane], s1 = red[1][lane], s2 = red[2][lane], s3 = red[3][lane];
        const float bc = bias[16 * ot + n];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = ((s0[r] + s1[r]) + (s2[r] + s3[r])) * inv_scale + bc;
    }
    return v;
}

template <int CIN, int GM>
__device__ __forceinline__ void bo_b1_fetch_weights(bo_f32x4 (&a)[9][GM], const bo_f32x4 *__restrict__ w, int ot, int wave, int lane) {
    constexpr int G = CIN / 64;
    const bo_f32x4 *wl = w + ((size_t)ot * 9 * (CIN / 16) + wave * G) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int g = 0; g < G; g++) a[tap][g] = wl[(tap * (CIN / 16) + g) * 64];
}

enum { BO_B1_F32 = 0, BO_B1_SPLIT = 1 };   // the matrix pipe a tile multiplies on: fp32 MFMA (exact float32 products) or fp16 MFMA with (hi, lo) operand pairs
template <int C, int MODE, int PREC>
__global__ void __launch_bounds__(256)
bo_k_tower_b1(bo_b1_args A) {
    constexpr int TILES = (C / 16) * 4, CM = (C > 128 ? C : 128), CQM = CM / 4, GM = CM / 64, SLAB = 40;
    __shared__ float Xs[4][CQM * SLAB];
    __shared__ bo_f32x4 red[4][64];
    __shared__ float mean[C], hid[16], gate16[16], part[256];
    __shared__ int ok_lds;
    // Workgroups are dealt round-robin over the 8 XCDs (observed, for speed only): with ot in the LOW bits the four position tiles of a
    // channel tile (which read the same 147 KB of weights per layer) share an XCD, so the weights cross the fabric once per channel
    // tile instead of four times (the 100 MB stream of an evaluation was 400 MB requested; profiles/r04_b1_tower.md).
    const int ot = blockIdx.x % (C / 16), pt = blockIdx.x / (C / 16), b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    unsigned *ctr = A.sync + b, *status = A.sync + 2 * A.MB;
    // what this board's counter stood at when the launch began: stable until the launch's last arriver rewrites it, long after every
    // workgroup has read it here
    const unsigned base = __hip_atomic_load(BO_GU32(A.sync + A.MB + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t plane = (size_t)C * 64;
    float *buf[3] = {A.bufs + (size_t)b * plane, A.bufs + ((size_t)A.B + b) * plane, A.bufs + ((size_t)2 * A.B + b) * plane};
    float *pool = A.pool + (size_t)b * 4 * C;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A.bufs, 0, 0x7fffffff, 0x00020000);  // the activation buffers (3 x B x C x 64 floats)
    for (int i = tid; i < 4 * CQM * SLAB; i += 256) (&Xs[0][0])[i] = 0.0f;  // halo columns and rows outside the board are never written again
    __syncthreads();

    bo_f32x4 a[9][GM];          // this wave's weight fragments of the layer at hand (input layer: 128 / 64 groups, then C / 64)
    const bo_b1_layer *L = A.layers;
    bo_b1_fetch_weights<128, GM>(a, PREC == BO_B1_SPLIT ? L[0].w_split : L[0].w, ot, wave, lane);
    bool overflow = false;
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tm[3] = {0, 0, 0};
    unsigned long long *tmp = A.prof ? tm : nullptr;
    unsigned phase = 0;         // hand-offs completed so far: the counter reaches phase * TILES when every tile of that phase is stored
    int cur = -1;               // buffer that holds the block input (-1: the kernel's input planes)
    for (int l = 0; l < A.n_layers; l++) {
        const bo_b1_layer ly = L[l];
        const bool first_of_block = (l >= 1) && ((l - 1) % 2 == 0);  // layers 1, 3, 5 ...: conv1 of a block; 2, 4, ...: conv2
        // buffers: input layer x -> buf0; a block reads cur, conv1 -> (cur + 1) % 3, conv2 -> (cur + 2) % 3 (+ residual from cur)
        const float *xin;
        float *xout;
        const float *res = nullptr;
        if (l == 0) { xin = A.x + (size_t)b * 120 * 64; xout = buf[0]; }
        else if (first_of_block) { xin = buf[cur]; xout = buf[(cur + 1) % 3]; }
        else { xin = buf[(cur + 1) % 3]; xout = buf[(cur + 2) % 3]; res = buf[cur]; }
        if (l == A.n_layers - 1) xout = A.y + (size_t)b * plane;
        const unsigned long long s0 = tmp ? __builtin_amdgcn_s_memtime() : 0ull;
        if (l > 0) {  // the previous layer's tiles: every workgroup of this board has arrived `phase` times
            if (!bo_b1_wait<MODE>(ctr, status, base + phase * TILES, 1u + phase, tid, &ok_lds)) return;
        }
        const unsigned long long s1 = tmp ? __builtin_amdgcn_s_memtime() : 0ull;
        const size_t o = (size_t)(16 * ot + n) * 64 + 16 * pt + 4 * kq;  // this lane's four squares of its channel (wave 0's epilogue)
        bo_f32x4 rres = {0, 0, 0, 0};  // the residual tile (the block input, handed over two phases ago): requested with the slab, used in the epilogue
        if (wave == 0 && ly.mode != 0) rres = bo_b1_load16<MODE>(rsrc, A.bufs, (size_t)(res - A.bufs) + o);
        const bool more = l + 1 < A.n_layers;
        const bo_f32x4 *wnext_all = more ? (PREC == BO_B1_SPLIT ? L[l + 1].w_split : L[l + 1].w) : nullptr;
        bo_f32x4 v;
        if constexpr (PREC == BO_B1_SPLIT) {
            if (l == 0) v = bo_b1_conv_split<128, GM, false, MODE, C>(xin, rsrc, A.bufs, ly.cin_x, a, &Xs[wave][0], red, ly.bias, ly.inv_scale, ot, pt, wave, lane, overflow, tmp);
            else v = bo_b1_conv_split<C, GM, true, MODE, C>(xin, rsrc, A.bufs, C, a, &Xs[wave][0], red, ly.bias, ly.inv_scale, ot, pt, wave, lane, overflow, tmp);
        } else {
            if (l == 0) v = bo_b1_conv<128, GM, false, MODE, C>(xin, rsrc, A.bufs, ly.cin_x, a, &Xs[wave][0], red, ly.bias, ot, pt, wave, lane, tmp);
            else v = bo_b1_conv<C, GM, true, MODE, C>(xin, rsrc, A.bufs, C, a, &Xs[wave][0], red, ly.bias, ot, pt, wave, lane, tmp);
        }
        // the next layer's weights: requested now (waves 1-3; wave 0 after it has signalled: its drain would wait for them), they arrive
        // while the tile is stored and the hand-off completes.  A wave keeps ~16 of these 1-KB loads in flight (~6.5 B/clk per wave, 26
        // per CU: 5 200 clocks per layer at 256 filters).  Measured and dropped (profiles/r04_b1_tower.md): refilling the fragment
        // registers tap by tap under the MFMAs -- the loads stall the issuing wave, the matrix phase grew by more than the fetch shrank.
        if (more && wave != 0) bo_b1_fetch_weights<C, GM>(a, wnext_all, ot, wave, lane);
        if (ly.mode != 2) {
            if (wave == 0) {
                if (ly.mode == 1) {
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] += rres[q];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = v[q] > 0.0f ? v[q] : 0.0f;
                bo_b1_store16_sc1(xout + o, v);
                if (more) bo_b1_arrive(ctr, lane);
                else bo_b1_arrive_last(ctr, A.sync + A.MB + b, base + (phase + 1u) * TILES, lane);
                if (more) bo_b1_fetch_weights<C, GM>(a, wnext_all, ot, wave, lane);
            }
            phase++;
        } else {
            // SE block: partial channel sums of this tile -> pool[pt][channel]; the un-gated tile stays in wave 0's registers
            if (wave == 0) {
                float s = (v[0] + v[1]) + (v[2] + v[3]);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (kq == 0) bo_b1_store4_sc1(pool + (size_t)pt * C + 16 * ot + n, s);
                bo_b1_arrive(ctr, lane);
            }
            phase++;
            // FC1 weights of this thread: 16 threads per hidden unit, each a slice of C/16 channels (requested before the wait)
            constexpr int PER = C / 16;
            const int h = tid >> 4, sl = tid & 15;
            float w1r[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) w1r[i] = h < ly.se_h ? ly.se_w1[(size_t)h * C + sl * PER + i] : 0.0f;
            if (!bo_b1_wait<MODE>(ctr, status, base + phase * TILES, 1u + phase, tid, &ok_lds)) return;
            if (tid < C)  // AdaptiveAvgPool2d(1)
                mean[tid] = ((bo_b1_load4<MODE>(pool + tid) + bo_b1_load4<MODE>(pool + C + tid)) + (bo_b1_load4<MODE>(pool + 2 * C + tid) + bo_b1_load4<MODE>(pool + 3 * C + tid))) * (1.0f / 64.0f);
            __syncthreads();
            {   // Linear(C, C/r, bias=False) + ReLU
                float acc1 = 0.0f;
#pragma unroll
                for (int i = 0; i < PER; i++) acc1 += w1r[i] * mean[sl * PER + i];
                part[tid] = acc1;
                __syncthreads();
                if (sl == 0 && h < ly.se_h) {
                    float sum = 0.0f;
                    for (int i = 0; i < 16; i++) sum += part[(tid & ~15) + i];
                    hid[h] = sum > 0.0f ? sum : 0.0f;
                }
                __syncthreads();
            }
            if (tid < 16) {  // Linear(C/r, C, bias=False) + Sigmoid for this workgroup's channels
                const int c = 16 * ot + tid;
                float g = 0.0f;
                for (int j = 0; j < ly.se_h; j++) g += ly.se_w2[(size_t)c * ly.se_h + j] * hid[j];
                gate16[tid] = 1.0f / (1.0f + expf(-g));
            }
            __syncthreads();
            if (wave == 0) {
                const float g = gate16[n];
#pragma unroll
                for (int q = 0; q < 4; q++) { v[q] = v[q] * g + rres[q]; v[q] = v[q] > 0.0f ? v[q] : 0.0f; }
                bo_b1_store16_sc1(xout + o, v);
                if (more) bo_b1_arrive(ctr, lane);
                else bo_b1_arrive_last(ctr, A.sync + A.MB + b, base + (phase + 1u) * TILES, lane);
                if (more) bo_b1_fetch_weights<C, GM>(a, wnext_all, ot, wave, lane);
            }
            phase++;
        }
        if (tmp) {
            const unsigned long long s5 = __builtin_amdgcn_s_memtime();
            tsum[0] += s1 - s0; tsum[1] += tm[0] - s1; tsum[2] += tm[1] - tm[0]; tsum[3] += tm[2] - tm[1]; tsum[4] += s5 - tm[2]; tsum[5] += 1;
        }
        if (l == 0) cur = 0;
        else if (!first_of_block) cur = (cur + 2) % 3;
    }
    if (tmp && lane == 0) {
        unsigned long long *o = A.prof + (((size_t)b * TILES + blockIdx.x) * 4 + wave) * 8;
        for (int k = 0; k < 6; k++) o[k] = tsum[k];
    }
    if (PREC == BO_B1_SPLIT && overflow)  // an activation left the fp16 range: the evaluation is wrong (bo_nn_b1_status)
        __hip_atomic_fetch_or(BO_GU32(A.sync + 2 * A.MB + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
