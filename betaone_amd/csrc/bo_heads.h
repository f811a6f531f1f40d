// betaone_amd/csrc/bo_heads.h -- everything behind the tower's head convolutions in two launches (gfx950).
//
// PolicyValueNet.forward after the two 1x1 head convolutions (/root/reference/network.py:186-197) and the softmax the
// search applies to the logits (/root/reference/mcts.py:185,287):
//     logits = policy_fc(p)            [B,128]  x [128,4672]           p = ReLU'd policy planes, flattened
//     probs  = softmax(logits, dim=1)
//     value  = tanh(value_fc2(relu(value_fc1(v))))   [B,2048] x [2048,256], then 256 -> 1;  v = ReLU'd value planes
// As library calls this is two small GEMMs on forked streams, a softmax and a tail kernel: ~25 us of kernels that are launch-
// and latency-bound (0.6 GFLOP in total) plus ~15 us of dependency gaps between graph nodes, behind every one of the ten
// evaluations of a ply.  Here:
//   bo_k_heads_tiles   workgroups 0 .. NP-1   logits tile [128 boards x 32 outputs] on v_mfma_f32_16x16x4_f32 (K = 128): every
//                                             fragment of a wave is requested before its first MFMA (one memory round trip)
//                      workgroups NP ..       value_fc1 partial tile [64 boards x 64 hidden x 128 of K = 2048]: both operand
//                                             tiles are staged through LDS with coalesced loads; partial sums go to scratch
//   bo_k_heads_rows    one board per workgroup: softmax of its logits row (the row passes through registers once) and
//                      value = tanh(value_fc2(relu(sum of the 16 K chunks in a fixed order + bias)))
// (A single launch with a device-wide barrier between the two phases was measured first: what crosses the barrier has to be
// written through / fetched past the XCDs' private L2s, or every workgroup has to write back and invalidate its L2; either way
// the tiles' stores took longer than the kernel boundary costs -- profiles/r02_heads_probe.md.)
// Both products are computed transposed (M = outputs, N = boards), so that a lane's accumulator is four consecutive outputs of
// one board: 16-byte stores.  Operands need no packing: a lane's fragments of four consecutive K-steps are one float4 of a
// row of the activations / of the Linear weight (any partition of K into groups of four is a valid K-step as long as both
// operands use the same one).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_tower.h"

#define BO_HEADS_NA 4672
#define BO_HEADS_KP 128
#define BO_HEADS_KV 2048
#define BO_HEADS_NH 256
#define BO_HEADS_KS 16        // K chunks of value_fc1 (partial sums reduced by the rows kernel, in a fixed order)
#define BO_HEADS_PITCH 132    // floats per LDS tile row: 128 + 4 (conflict-free 16-byte fragment reads)
#define BO_HEADS_PROWS 128    // boards per logits tile

struct bo_heads_args {
    const void *p, *v;                        // [B,128], [B,2048]: float32, or float16 behind the fp16 tower (bo_tower_h.h)
    const float *wp, *bp, *w1, *b1, *w2, *b2;  // policy_fc [4672,128]+[4672]; value_fc1 [256,2048]+[256]; value_fc2 [256]+[1]
    float *policy_out, *value_out;            // [B,4672] (probabilities if softmax != 0, else logits), [B]
    float *vpart;                             // scratch [16 K chunks][B][256]: partial sums of value_fc1
    int B, softmax;
};

// four consecutive activations as float32 from a float32 or float16 array (element index e, a multiple of 4)
template <bool HALF>
__device__ __forceinline__ bo_f32x4 bo_heads_load4(const void *base, size_t e) {
    if constexpr (HALF) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 h = *reinterpret_cast<const h4 *>(reinterpret_cast<const _Float16 *>(base) + e);
        return bo_f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        return *reinterpret_cast<const bo_f32x4 *>(reinterpret_cast<const float *>(base) + e);
    }
}

template <bool HALF>
__global__ void __launch_bounds__(256)
bo_k_heads_tiles(bo_heads_args a) {
    __shared__ __attribute__((aligned(16))) float tileA[64 * BO_HEADS_PITCH], tileW[64 * BO_HEADS_PITCH];  // value tiles [64][128 + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, i = lane & 15;
    const int B = a.B, NPT = BO_HEADS_NA / 32, n_policy = ((B + BO_HEADS_PROWS - 1) / BO_HEADS_PROWS) * NPT;
    const int wg = (int)blockIdx.x;
    if (wg < n_policy) {
        // ---- logits tile: boards 128*rb + 32*wave + [0,32) (N), outputs 32*ct + [0,32) (M) ----
        const int rb = wg / NPT, ct = wg - rb * NPT, r0 = BO_HEADS_PROWS * rb + 32 * wave, c0 = 32 * ct;
        const bo_f32x4 *w4 = reinterpret_cast<const bo_f32x4 *>(a.wp);
        constexpr int PG = BO_HEADS_KP / 16;
        bo_f32x4 fp[PG][2], fw[PG][2];
#pragma unroll
        for (int t = 0; t < PG; t++) {
            const int kk = 16 * t + 4 * kq;
#pragma unroll
            for (int c = 0; c < 2; c++) fw[t][c] = w4[((size_t)(c0 + 16 * c + i) * BO_HEADS_KP + kk) >> 2];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int row = r0 + 16 * rt + i;
                fp[t][rt] = row < B ? bo_heads_load4<HALF>(a.p, (size_t)row * BO_HEADS_KP + kk) : bo_f32x4{0, 0, 0, 0};
            }
        }
        bo_f32x4 bias[2];
#pragma unroll
        for (int c = 0; c < 2; c++) bias[c] = *reinterpret_cast<const bo_f32x4 *>(a.bp + c0 + 16 * c + 4 * kq);
        __builtin_amdgcn_sched_barrier(0);
        bo_f32x4 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) acc[c][rt] = bo_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < PG; t++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[t][c][e], fp[t][rt][e], acc[c][rt], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = c0 + 16 * c + 4 * kq;  // this lane: outputs col .. col+3 of board r0 + 16*rt + i
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int row = r0 + 16 * rt + i;
                if (row < B)
                    *reinterpret_cast<bo_f32x4 *>(a.policy_out + (size_t)row * BO_HEADS_NA + col) =
                        bo_f32x4{acc[c][rt][0] + bias[c][0], acc[c][rt][1] + bias[c][1], acc[c][rt][2] + bias[c][2], acc[c][rt][3] + bias[c][3]};
            }
        }
    } else {
        // ---- value_fc1 partial tile: boards 64*rbv + [0,64), hidden units 64*ht + [0,64), K chunk 128*ks + [0,128).
        //      Both operand tiles are contiguous 512-byte row segments: fetched with whole-wave coalesced loads into LDS (a
        //      fragment-order fetch of rows 8 KB apart serialises on a few L2 channels), fragments come from LDS. ----
        const int vt = wg - n_policy, ks = vt & (BO_HEADS_KS - 1), ht = (vt >> 4) & 3, rbv = vt >> 6;
        const int r0 = 64 * rbv, h0 = 64 * ht, k0 = (BO_HEADS_KV / BO_HEADS_KS) * ks;
        const bo_f32x4 *w4 = reinterpret_cast<const bo_f32x4 *>(a.w1);
        bo_f32x4 ga[8], gw[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, row = idx >> 5, c4 = idx & 31;  // 32 lanes per 512-byte row segment
            ga[q] = r0 + row < B ? bo_heads_load4<HALF>(a.v, ((size_t)(r0 + row)) * BO_HEADS_KV + k0 + 4 * c4) : bo_f32x4{0, 0, 0, 0};
            gw[q] = w4[(((size_t)(h0 + row)) * BO_HEADS_KV + k0) / 4 + c4];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, row = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<bo_f32x4 *>(&tileA[row * BO_HEADS_PITCH + 4 * c4]) = ga[q];
            *reinterpret_cast<bo_f32x4 *>(&tileW[row * BO_HEADS_PITCH + 4 * c4]) = gw[q];
        }
        __syncthreads();
        // wave: hidden units 32*(wave & 1) + [0,32) (M), boards 32*(wave >> 1) + [0,32) (N)
        const int hb = 32 * (wave & 1), bb = 32 * (wave >> 1);
        bo_f32x4 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) acc[c][rt] = bo_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < BO_HEADS_KV / BO_HEADS_KS / 16; t++) {
            bo_f32x4 fw[2], fv[2];
#pragma unroll
            for (int c = 0; c < 2; c++) fw[c] = *reinterpret_cast<const bo_f32x4 *>(&tileW[(hb + 16 * c + i) * BO_HEADS_PITCH + 16 * t + 4 * kq]);
#pragma unroll
            for (int rt = 0; rt < 2; rt++) fv[rt] = *reinterpret_cast<const bo_f32x4 *>(&tileA[(bb + 16 * rt + i) * BO_HEADS_PITCH + 16 * t + 4 * kq]);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[c][e], fv[rt][e], acc[c][rt], 0, 0, 0);
        }
        // partial sums [ks][board][256 hidden]: this lane holds hidden h0 + hb + 16c + 4kq + [0,4) of board r0 + bb + 16rt + i
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int row = r0 + bb + 16 * rt + i;
                if (row < B) *reinterpret_cast<bo_f32x4 *>(a.vpart + ((size_t)ks * B + row) * BO_HEADS_NH + h0 + hb + 16 * c + 4 * kq) = acc[c][rt];
            }
    }
}

// one board per workgroup (grid = B): the softmax of its logits row and its value
extern "C" __global__ void __launch_bounds__(256)
bo_k_heads_rows(bo_heads_args a) {
    __shared__ float sred[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = a.B, row = (int)blockIdx.x;
    // value: the K chunks' partial sums in a fixed order, bias, ReLU, value_fc2, tanh (requested first: it overlaps the row)
    float part[BO_HEADS_KS];
#pragma unroll
    for (int ks = 0; ks < BO_HEADS_KS; ks++) part[ks] = a.vpart[((size_t)ks * B + row) * BO_HEADS_NH + tid];
    bo_f32x4 *r4 = reinterpret_cast<bo_f32x4 *>(a.policy_out + (size_t)row * BO_HEADS_NA);
    constexpr int N4 = BO_HEADS_NA / 4, IT = (N4 + 255) / 256;  // 1168 float4, 5 per thread
    const float ninf = -__builtin_inff();
    bo_f32x4 x[IT];
    float mx = ninf;
    if (a.softmax) {
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int k = tid + 256 * u;
            x[u] = k < N4 ? r4[k] : bo_f32x4{ninf, ninf, ninf, ninf};
        }
    }
    float hsum = 0.0f;
#pragma unroll
    for (int ks = 0; ks < BO_HEADS_KS; ks++) hsum += part[ks];
    hsum += a.b1[tid];
    hsum = (hsum > 0.0f ? hsum : 0.0f) * a.w2[tid];
    if (a.softmax) {
#pragma unroll
        for (int u = 0; u < IT; u++) mx = fmaxf(mx, fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3])));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, m, 64)); hsum += __shfl_xor(hsum, m, 64); }
    if (lane == 0) { sred[wave] = mx; sred[4 + wave] = hsum; }
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    if (tid == 0) a.value_out[row] = tanhf(((sred[4] + sred[5]) + (sred[6] + sred[7])) + a.b2[0]);
    if (a.softmax) {
        float sum = 0.0f;
#pragma unroll
        for (int u = 0; u < IT; u++) {
            if (tid + 256 * u < N4) {
#pragma unroll
                for (int e = 0; e < 4; e++) { x[u][e] = expf(x[u][e] - mx); sum += x[u][e]; }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
        __syncthreads();
        if (lane == 0) sred[wave] = sum;
        __syncthreads();
        sum = (sred[0] + sred[1]) + (sred[2] + sred[3]);
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int k = tid + 256 * u;
            if (k < N4) r4[k] = bo_f32x4{x[u][0] / sum, x[u][1] / sum, x[u][2] / sum, x[u][3] / sum};
        }
    }
}
#endif
