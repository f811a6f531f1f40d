// betaone_amd/csrc/bo_heads.h -- everything behind the tower's head convolutions in ONE launch (gfx950).
//
// PolicyValueNet.forward after the two 1x1 head convolutions (/root/reference/network.py:186-197) and the softmax the
// search applies to the logits (/root/reference/mcts.py:185,287):
//     logits = policy_fc(p)            [B,128]  x [128,4672]           p = ReLU'd policy planes, flattened
//     probs  = softmax(logits, dim=1)
//     value  = tanh(value_fc2(relu(value_fc1(v))))   [B,2048] x [2048,256], then 256 -> 1;  v = ReLU'd value planes
// As library calls this is two small GEMMs on forked streams, a softmax and a tail kernel: ~25 us of kernels that are launch-
// and latency-bound (0.6 GFLOP in total) plus ~15 us of dependency gaps between graph nodes, behind every one of the ten
// evaluations of a ply.  Here it is one kernel of two phases separated by a device-wide barrier:
//   phase A   workgroups 0 .. NP-1      logits tile [256 boards x 32 outputs] on v_mfma_f32_16x16x4_f32 (K = 128)
//             workgroups NP ..          value_fc1 tile [32 boards x 16 hidden], the four waves split K = 2048 and reduce
//                                       through LDS; bias, ReLU and this tile's share of value_fc2 (a partial dot product)
//   barrier   every workgroup is resident (<= 2 per CU at 512 boards), so an atomic counter suffices; the spin is bounded
//   phase B   softmax of the logits rows (one row per workgroup at a time, the row passes through registers once);
//             value = tanh(sum of the 16 partials + bias)
// Operands need no packing: a lane's A / B fragments of four consecutive K-steps are one float4 of a row of p / of a row of
// the Linear weight (any partition of K into groups of four is a valid K-step as long as A and B use the same one).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_tower.h"

#define BO_HEADS_NA 4672
#define BO_HEADS_KP 128
#define BO_HEADS_KV 2048
#define BO_HEADS_NH 256
#define BO_HEADS_MAX_B 512

struct bo_heads_args {
    const float *p, *v;                       // [B,128], [B,2048]
    const float *wp, *bp, *w1, *b1, *w2, *b2;  // policy_fc [4672,128]+[4672]; value_fc1 [256,2048]+[256]; value_fc2 [256]+[1]
    float *policy_out, *value_out;            // [B,4672] (probabilities if softmax != 0, else logits), [B]
    float *vpart;                             // scratch [B,16]
    unsigned *bar;                            // [4] zero before the first launch: arrivals, departures, error flag, unused
    int B, softmax;
};

__device__ __forceinline__ void bo_heads_grid_barrier(unsigned *bar, unsigned n) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(&bar[0], 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { bar[2] = 1u; break; }  // (never in practice: all workgroups are resident) no hang, an error flag
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

extern "C" __global__ void __launch_bounds__(256)
bo_k_heads(bo_heads_args a) {
    __shared__ bo_f32x4 red[4][2][64];
    __shared__ float sred[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, i = lane & 15;
    const int B = a.B, NPT = BO_HEADS_NA / 32, n_policy = ((B + 255) / 256) * NPT;
    const int wg = (int)blockIdx.x;
    if (wg < n_policy) {
        // ---- logits tile: rows 256*rb + 64*wave + [0,64), columns 32*ct + [0,32) ----
        const int rb = wg / NPT, ct = wg - rb * NPT, r0 = 256 * rb + 64 * wave, c0 = 32 * ct;
        bo_f32x4 acc[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; rt++)
#pragma unroll
            for (int c = 0; c < 2; c++) acc[rt][c] = bo_f32x4{0, 0, 0, 0};
        const bo_f32x4 *p4 = reinterpret_cast<const bo_f32x4 *>(a.p), *w4 = reinterpret_cast<const bo_f32x4 *>(a.wp);
        // the fragments of half the K range are requested before their first MFMA (24 float4 per lane, two memory round trips
        // per tile; fetched per K-group the tile was a chain of eight dependent round trips).  Register budget: under 256 per
        // lane, so that two workgroups share a CU -- the device-wide barrier needs every workgroup resident.
        constexpr int PG = BO_HEADS_KP / 16 / 2;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            bo_f32x4 fa[PG][4], fb[PG][2];
#pragma unroll
            for (int t = 0; t < PG; t++) {
                const int kk = 16 * (PG * half + t) + 4 * kq;
#pragma unroll
                for (int rt = 0; rt < 4; rt++) {
                    const int row = r0 + 16 * rt + i;
                    fa[t][rt] = row < B ? p4[((size_t)row * BO_HEADS_KP + kk) >> 2] : bo_f32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int c = 0; c < 2; c++) fb[t][c] = w4[((size_t)(c0 + 16 * c + i) * BO_HEADS_KP + kk) >> 2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < PG; t++)
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int rt = 0; rt < 4; rt++)
#pragma unroll
                        for (int c = 0; c < 2; c++) acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t][rt][e], fb[t][c][e], acc[rt][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = c0 + 16 * c + i;
            const float bc = a.bp[col];
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = r0 + 16 * rt + 4 * kq + r;
                    if (row < B) a.policy_out[(size_t)row * BO_HEADS_NA + col] = acc[rt][c][r] + bc;
                }
        }
    } else {
        // ---- value_fc1 tile: rows 32*rbv + [0,32), hidden units 16*ht + [0,16); wave = quarter of K ----
        const int vt = wg - n_policy, rbv = vt >> 4, ht = vt & 15, r0 = 32 * rbv, k0 = (BO_HEADS_KV / 4) * wave;
        bo_f32x4 acc[2] = {bo_f32x4{0, 0, 0, 0}, bo_f32x4{0, 0, 0, 0}};
        const bo_f32x4 *v4 = reinterpret_cast<const bo_f32x4 *>(a.v), *w4 = reinterpret_cast<const bo_f32x4 *>(a.w1);
        constexpr int VG = 8;  // groups of 16 K per round trip: 24 float4 per lane in flight, four round trips per wave
        for (int t0 = 0; t0 < BO_HEADS_KV / 4 / 16; t0 += VG) {
            bo_f32x4 fa[VG][2], fb[VG];
#pragma unroll
            for (int u = 0; u < VG; u++) {
                const int kk = k0 + 16 * (t0 + u) + 4 * kq;
#pragma unroll
                for (int rt = 0; rt < 2; rt++) {
                    const int row = r0 + 16 * rt + i;
                    fa[u][rt] = row < B ? v4[((size_t)row * BO_HEADS_KV + kk) >> 2] : bo_f32x4{0, 0, 0, 0};
                }
                fb[u] = w4[((size_t)(16 * ht + i) * BO_HEADS_KV + kk) >> 2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < VG; u++)
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][rt][e], fb[u][e], acc[rt], 0, 0, 0);
        }
        red[wave][0][lane] = acc[0];
        red[wave][1][lane] = acc[1];
        __syncthreads();
        if (wave == 0) {
            const int h = 16 * ht + i;
            const float b1 = a.b1[h], w2 = a.w2[h];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const bo_f32x4 s0 = red[0][rt][lane], s1 = red[1][rt][lane], s2 = red[2][rt][lane], s3 = red[3][rt][lane];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float hv = ((s0[r] + s1[r]) + (s2[r] + s3[r])) + b1;  // value_fc1 + bias
                    hv = hv > 0.0f ? hv : 0.0f;                            // ReLU
                    float part = hv * w2;                                  // this hidden unit's term of value_fc2
                    part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
                    part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);  // over the tile's 16 hidden units (lanes i)
                    const int row = r0 + 16 * rt + 4 * kq + r;
                    if (i == 0 && row < B) a.vpart[(size_t)row * 16 + ht] = part;
                }
            }
        }
    }

    bo_heads_grid_barrier(a.bar, gridDim.x);

    // ---- phase B: softmax rows (round-robin over the workgroups), then the value tail ----
    if (a.softmax) {
        for (int row = wg; row < B; row += (int)gridDim.x) {
            bo_f32x4 *r4 = reinterpret_cast<bo_f32x4 *>(a.policy_out + (size_t)row * BO_HEADS_NA);
            constexpr int N4 = BO_HEADS_NA / 4, IT = (N4 + 255) / 256;  // 1168 float4, 5 per thread
            const float ninf = -__builtin_inff();
            bo_f32x4 x[IT];
            float mx = ninf;
#pragma unroll
            for (int u = 0; u < IT; u++) {
                const int k = tid + 256 * u;
                x[u] = k < N4 ? r4[k] : bo_f32x4{ninf, ninf, ninf, ninf};
                mx = fmaxf(mx, fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3])));
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            __syncthreads();  // (sred of the previous row has been read)
            if (lane == 0) sred[wave] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
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
            if (lane == 0) sred[4 + wave] = sum;
            __syncthreads();
            sum = (sred[4] + sred[5]) + (sred[6] + sred[7]);
#pragma unroll
            for (int u = 0; u < IT; u++) {
                const int k = tid + 256 * u;
                if (k < N4) r4[k] = bo_f32x4{x[u][0] / sum, x[u][1] / sum, x[u][2] / sum, x[u][3] / sum};
            }
        }
    }
    if (wg == (int)gridDim.x - 1) {
        const float b2 = a.b2[0];
        for (int row = tid; row < B; row += 256) {
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < 16; t++) s += a.vpart[(size_t)row * 16 + t];
            a.value_out[row] = tanhf(s + b2);
        }
    }
    // the last workgroup to leave re-arms the barrier for the next launch
    __syncthreads();
    if (tid == 0) {
        if (atomicAdd(&a.bar[1], 1u) == gridDim.x - 1) { a.bar[0] = 0u; a.bar[1] = 0u; }
    }
}
#endif
