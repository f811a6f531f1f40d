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
//     to the board's arrival counter (agent scope); a consumer polls that ONE word relaxed, runs ONE agent-scope acquire, waits for it,
//     joins the workgroup barrier, then every wave reads its slab with plain 16-byte loads (cdna_hip_programming.md, Guideline 16: R1
//     producer, "Consumer, always" form).  Per-XCD L2s are not coherent and HIP promises nothing about placement: nothing here depends
//     on which XCD a workgroup runs on;
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

struct bo_b1_layer {        // one 3x3 convolution of the tower (device table)
    const bo_f32x4 *w;      // fused_net.pack_conv_weight_small: [C/16][tap 9][cin/16][64][4]
    const float *bias;      // [C]
    const float *se_w1;     // [H][C]  (mode 2)
    const float *se_w2;     // [C][H]
    int cin, cin_x;         // K channels the weights are packed for (128 for the input layer); channels present in the input
    int mode;               // 0: relu(conv + bias)   1: relu(conv + bias + residual)   2: relu((conv + bias) * se_gate + residual)
    int se_h;
};

struct bo_b1_args {
    const float *x;         // [B][120][64] input planes
    float *y;               // [B][C][64] tower output
    float *bufs;            // [3][B][C][64] activations between layers
    float *pool;            // [B][4][C] partial channel sums of an SE layer (one row per position tile)
    unsigned *sync;         // [0..B): arrival counters; [B]: status word (0 ok, else 1 + phase of the wait that gave up).  Zeroed before every launch.
    const bo_b1_layer *layers;
    int n_layers, B;
};

typedef __attribute__((address_space(1))) unsigned bo_gu32;
#define BO_GU32(p) ((bo_gu32 *)(p))   // a generic pointer known to be global: GLOBAL (not flat) agent-scope accesses

__device__ __forceinline__ void bo_b1_store16_sc1(float *p, bo_f32x4 v) {  // write-through: the bytes leave this XCD's L2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bo_b1_store4_sc1(float *p, float v) {
    __hip_atomic_store(BO_GU32(reinterpret_cast<unsigned *>(p)), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The storing wave (wave 0) has issued its write-through stores: drain them, then ONE lane signals.
__device__ __forceinline__ void bo_b1_arrive(unsigned *ctr, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(BO_GU32(ctr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every wave calls this; ONE lane (tid 64: wave 1, which stores nothing) polls the board's counter (relaxed, bounded), ONE agent acquire, its wait, the workgroup
// barrier; afterwards plain loads of the handed-off bytes are valid in every wave.  false: the wait gave up (status word written) or
// another workgroup did -- uniform over the workgroup.
__device__ __forceinline__ bool bo_b1_wait(unsigned *ctr, unsigned *status, unsigned target, unsigned code, int tid, int *ok_lds) {
    if (tid == 64) {
        bool ok = true;
        unsigned spins = 0;
        while (__hip_atomic_load(BO_GU32(ctr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > BO_B1_SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(BO_GU32(status), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (spins > BO_B1_SPIN_LIMIT) __hip_atomic_store(BO_GU32(status), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *ok_lds = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// One convolution of one tile: stage this wave's slab, its share of K on the matrix pipe, reduction over the four waves.
// Returns (in wave 0) conv + bias for squares 16*pt + 4*kq + [0,4) of channel 16*ot + n.
template <int CIN, int GM>
__device__ __forceinline__ bo_f32x4 bo_b1_conv(const float *__restrict__ xin, int cin_x, const bo_f32x4 (&a)[9][GM], float *Xw,
                                               bo_f32x4 (*red)[64], const float *__restrict__ bias, int ot, int pt, int wave, int lane) {
    constexpr int CQ = CIN / 4, G = CQ / 16, SLAB = 40, NST = CQ / 8;
    const int kq = lane >> 4, n = lane & 15;
    {   // slab: channels wave*CQ .. +CQ, board rows 2pt-1 .. 2pt+2 (32 contiguous floats per channel where the rows exist), halo columns stay zero
        bo_f32x4 st[NST];
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i >> 3, q = i & 7, row = 2 * pt - 1 + (q >> 1);
            const bool in = row >= 0 && row < 8 && wave * CQ + c < cin_x;
            st[u] = in ? *reinterpret_cast<const bo_f32x4 *>(xin + (size_t)(wave * CQ + c) * 64 + row * 8 + (q & 1) * 4) : bo_f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i >> 3, q = i & 7;
            float *d = Xw + c * SLAB + (q >> 1) * 10 + 1 + (q & 1) * 4;
            d[0] = st[u][0]; d[1] = st[u][1]; d[2] = st[u][2]; d[3] = st[u][3];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the slab is this wave's own (no workgroup barrier needed)
    __builtin_amdgcn_wave_barrier();
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
    red[wave][lane] = acc;
    __syncthreads();
    bo_f32x4 v = {0, 0, 0, 0};
    if (wave == 0) {
        const bo_f32x4 s0 = red[0][lane], s1 = red[1][lane], s2 = red[2][lane], s3 = red[3][lane];
        const float bc = bias[16 * ot + n];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = ((s0[r] + s1[r]) + (s2[r] + s3[r])) + bc;
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

template <int C>
__global__ void __launch_bounds__(256)
bo_k_tower_b1(bo_b1_args A) {
    constexpr int TILES = (C / 16) * 4, CM = (C > 128 ? C : 128), CQM = CM / 4, GM = CM / 64, SLAB = 40;
    __shared__ float Xs[4][CQM * SLAB];
    __shared__ bo_f32x4 red[4][64];
    __shared__ float mean[C], hid[16], gate16[16], part[256];
    __shared__ int ok_lds;
    const int ot = blockIdx.x >> 2, pt = blockIdx.x & 3, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    unsigned *ctr = A.sync + b, *status = A.sync + A.B;
    const size_t plane = (size_t)C * 64;
    float *buf[3] = {A.bufs + (size_t)b * plane, A.bufs + ((size_t)A.B + b) * plane, A.bufs + ((size_t)2 * A.B + b) * plane};
    float *pool = A.pool + (size_t)b * 4 * C;
    for (int i = tid; i < 4 * CQM * SLAB; i += 256) (&Xs[0][0])[i] = 0.0f;  // halo columns and rows outside the board are never written again
    __syncthreads();

    bo_f32x4 a[9][GM];          // this wave's weight fragments of the layer at hand (input layer: 128 / 64 groups, then C / 64)
    const bo_b1_layer *L = A.layers;
    bo_b1_fetch_weights<128, GM>(a, L[0].w, ot, wave, lane);
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
        if (l > 0) {  // the previous layer's tiles: every workgroup of this board has arrived `phase` times
            if (!bo_b1_wait(ctr, status, phase * TILES, 1u + phase, tid, &ok_lds)) return;
        }
        bo_f32x4 v;
        if (l == 0) v = bo_b1_conv<128, GM>(xin, ly.cin_x, a, &Xs[wave][0], red, ly.bias, ot, pt, wave, lane);
        else v = bo_b1_conv<C, GM>(xin, C, a, &Xs[wave][0], red, ly.bias, ot, pt, wave, lane);
        // the next layer's weights: requested now (waves 1-3; wave 0 after it has signalled: its drain would wait for them), they
        // arrive while the tile is stored and the hand-off completes
        const bool more = l + 1 < A.n_layers;
        if (more && wave != 0) bo_b1_fetch_weights<C, GM>(a, L[l + 1].w, ot, wave, lane);
        const size_t o = (size_t)(16 * ot + n) * 64 + 16 * pt + 4 * kq;
        if (ly.mode != 2) {
            if (wave == 0) {
                if (ly.mode == 1) {
                    const bo_f32x4 r = *reinterpret_cast<const bo_f32x4 *>(res + o);
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] += r[q];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = v[q] > 0.0f ? v[q] : 0.0f;
                bo_b1_store16_sc1(xout + o, v);
                bo_b1_arrive(ctr, lane);
                if (more) bo_b1_fetch_weights<C, GM>(a, L[l + 1].w, ot, wave, lane);
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
            if (!bo_b1_wait(ctr, status, phase * TILES, 1u + phase, tid, &ok_lds)) return;
            if (tid < C) mean[tid] = ((pool[tid] + pool[C + tid]) + (pool[2 * C + tid] + pool[3 * C + tid])) * (1.0f / 64.0f);  // AdaptiveAvgPool2d(1)
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
                const bo_f32x4 r = *reinterpret_cast<const bo_f32x4 *>(res + o);
                const float g = gate16[n];
#pragma unroll
                for (int q = 0; q < 4; q++) { v[q] = v[q] * g + r[q]; v[q] = v[q] > 0.0f ? v[q] : 0.0f; }
                bo_b1_store16_sc1(xout + o, v);
                bo_b1_arrive(ctr, lane);
                if (more) bo_b1_fetch_weights<C, GM>(a, L[l + 1].w, ot, wave, lane);
            }
            phase++;
        }
        if (l == 0) cur = 0;
        else if (!first_of_block) cur = (cur + 2) % 3;
    }
}
#endif
