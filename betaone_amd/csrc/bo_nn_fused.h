// betaone_amd/csrc/bo_nn_fused.h -- fused epilogues for the evaluate stage (NCHW float32, 8x8 boards).
//
// The residual tower of PolicyValueNet (/root/reference/network.py:48-118) is conv3x3 -> BN -> ReLU -> conv3x3 -> BN
// [-> SE gate] -> + skip -> ReLU.  With BatchNorm folded into the convolutions (eval mode) PyTorch-ROCm runs the
// convolutions in MIOpen's fp32 Winograd kernel and everything else as ~60 separate 5-10 us elementwise launches
// per forward (26 % of the forward at batch 256).  These two kernels replace them:
//   bo_k_bias_act      y = relu(x + bias[c] (+ residual))                in place, float4 per lane, HBM-bound
//   bo_k_se_residual   y = relu((x + bias[c]) * gate[b,c] + residual)    one workgroup per board: channel means,
//                      gate = sigmoid(W2 relu(W1 mean)) (network.py:33-45) and the scaled add in one pass over LDS
// Arithmetic order of the bias/residual path is the same as PyTorch's (x + b, then + r, then max 0), so those
// results are bit-identical; the SE mean is summed in a different order (differences ~1e-7, tolerance 1e-4).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>

extern "C" __global__ void __launch_bounds__(256)
bo_k_bias_act(float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ res, long n4, int C) {
    // n4 = B*C*64/4 float4 elements; 16 float4 per (b, c) plane
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i >> 4) % C);
        const float b = bias[c];
        float4 v = reinterpret_cast<float4 *>(x)[i];
        v.x += b; v.y += b; v.z += b; v.w += b;
        if (res) {
            const float4 r = reinterpret_cast<const float4 *>(res)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f;
        v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f;
        reinterpret_cast<float4 *>(x)[i] = v;
    }
}

#define BO_SE_MAX_C 256
#define BO_SE_MAX_H 32
extern "C" __global__ void __launch_bounds__(256)
bo_k_se_residual(float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ w1 /*[H][C]*/,
                 const float *__restrict__ w2 /*[C][H]*/, const float *__restrict__ res, int C, int H) {
    extern __shared__ float tile[];  // [C][65] padded rows: conflict-free per-channel sums
    __shared__ float mean[BO_SE_MAX_C], hid[BO_SE_MAX_H], gate[BO_SE_MAX_C], part[256];
    const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    float *xb = x + (size_t)b * C * 64;
    const float *rb = res + (size_t)b * C * 64;
    // FC1 is spread over all 256 threads: 16 threads per hidden unit, each a slice of the C channels (a batch of one board --
    // uci.py -- runs this kernel as a single workgroup, so its serial chains are the latency of the whole SE layer)
    const int sl = t & 15, per = (C + 15) >> 4;
    float w1r[BO_SE_MAX_C / 16];
#pragma unroll
    for (int i = 0; i < BO_SE_MAX_C / 16; i++) w1r[i] = 0.0f;
    for (int h0 = 0; h0 < H; h0 += 16) {  // (H <= 16: one pass)
        const int h = h0 + (t >> 4);
        if (h0 == 0 && h < H && t < 256)
#pragma unroll
            for (int i = 0; i < BO_SE_MAX_C / 16; i++)
                if (i < per && sl * per + i < C) w1r[i] = w1[h * C + sl * per + i];
    }
    for (int i = t; i < C * 64; i += nt) {  // coalesced load, + bias
        const int c = i >> 6, s = i & 63;
        tile[c * 65 + s] = xb[i] + bias[c];
    }
    __syncthreads();
    for (int c = t; c < C; c += nt) {  // AdaptiveAvgPool2d(1)
        float a = 0.0f;
        for (int s = 0; s < 64; s++) a += tile[c * 65 + s];
        mean[c] = a * (1.0f / 64.0f);
    }
    __syncthreads();
    for (int h0 = 0; h0 < H; h0 += 16) {  // Linear(C, C/r, bias=False) + ReLU
        const int h = h0 + (t >> 4);
        float a = 0.0f;
        if (h < H && t < 256) {
            if (h0 == 0) {
#pragma unroll
                for (int i = 0; i < BO_SE_MAX_C / 16; i++)
                    if (i < per && sl * per + i < C) a += w1r[i] * mean[sl * per + i];
            } else {
                for (int i = 0; i < per; i++)
                    if (sl * per + i < C) a += w1[h * C + sl * per + i] * mean[sl * per + i];
            }
        }
        if (t < 256) part[t] = a;
        __syncthreads();
        if (sl == 0 && h < H && t < 256) {
            float sum = 0.0f;
            for (int i = 0; i < 16; i++) sum += part[(t & ~15) + i];
            hid[h] = sum > 0.0f ? sum : 0.0f;
        }
        __syncthreads();
    }
    for (int c = t; c < C; c += nt) {  // Linear(C/r, C, bias=False) + Sigmoid
        float a = 0.0f;
        for (int j = 0; j < H; j++) a += w2[c * H + j] * hid[j];
        gate[c] = 1.0f / (1.0f + expf(-a));
    }
    __syncthreads();
    for (int i = t; i < C * 64; i += nt) {  // scale, skip connection, ReLU
        const int c = i >> 6, s = i & 63;
        const float v = tile[c * 65 + s] * gate[c] + rb[i];
        xb[i] = v > 0.0f ? v : 0.0f;
    }
}

// value tail: out[b] = tanh(w . h[b] + bias) (value_fc2 + tanh, /root/reference/network.py:116-118,197); one wave per board
extern "C" __global__ void __launch_bounds__(256)
bo_k_value_tail(const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ out, int B, int H) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float a = 0.0f;
    for (int i = lane; i < H; i += 64) a += h[(size_t)b * H + i] * w[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    if (lane == 0) out[b] = tanhf(a + bias[0]);
}
#endif
