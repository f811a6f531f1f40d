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

// Small-batch form of the SE epilogue (uci.py analyses ONE position: bo_k_se_residual would run as a single workgroup whose
// 64 dependent load iterations per thread are the latency of the whole layer, 30 us).  Here a board's layer is cut into C/16
// workgroups of 16 channels.  Every workgroup pools ALL channels itself (thread = channel, its 64 squares as 16 float4 loads
// issued at once; 64 KB from L2, cheaper than a rendezvous between workgroups), computes the hidden layer, then gates, adds
// the skip connection and applies ReLU for its own 16 channels.  x is only read; the result goes into `res` IN PLACE
// (res[i] = relu((x[i] + bias[c]) * gate[c] + res[i])): a workgroup reads and writes only its own elements of it.
extern "C" __global__ void __launch_bounds__(256)
bo_k_se_residual_small(const float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ w1 /*[H][C]*/,
                       const float *__restrict__ w2 /*[C][H]*/, float *__restrict__ res, int C, int H) {
    __shared__ float mean[BO_SE_MAX_C], hid[BO_SE_MAX_H], gate16[16], part[256];
    const int cg = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const float *xb = x + (size_t)b * C * 64;
    float *rb = res + (size_t)b * C * 64;
    const int sl = t & 15, per = (C + 15) >> 4;
    float w1r[BO_SE_MAX_C / 16];  // FC1: 16 threads per hidden unit, each a slice of the channels; requested before the pooling
#pragma unroll
    for (int i = 0; i < BO_SE_MAX_C / 16; i++) {
        const int h = t >> 4, c = sl * per + i;
        w1r[i] = (h < H && i < per && c < C) ? w1[h * C + c] : 0.0f;
    }
    if (t < C) {  // AdaptiveAvgPool2d(1) of channel t (conv output + bias)
        const float4 *p = reinterpret_cast<const float4 *>(xb + (size_t)t * 64);
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = p[i];
        const float bc = bias[t];
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) a += (v[i].x + bc) + (v[i].y + bc) + (v[i].z + bc) + (v[i].w + bc);
        mean[t] = a * (1.0f / 64.0f);
    }
    __syncthreads();
    {   // Linear(C, C/r, bias=False) + ReLU   (H <= 16)
        const int h = t >> 4;
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < BO_SE_MAX_C / 16; i++)
            if (i < per && sl * per + i < C) a += w1r[i] * mean[sl * per + i];
        part[t] = a;
        __syncthreads();
        if (sl == 0 && h < H) {
            float sum = 0.0f;
            for (int i = 0; i < 16; i++) sum += part[(t & ~15) + i];
            hid[h] = sum > 0.0f ? sum : 0.0f;
        }
        __syncthreads();
    }
    if (t < 16) {  // Linear(C/r, C, bias=False) + Sigmoid for this workgroup's channels
        const int c = cg * 16 + t;
        float a = 0.0f;
        for (int j = 0; j < H; j++) a += w2[c * H + j] * hid[j];
        gate16[t] = 1.0f / (1.0f + expf(-a));
    }
    __syncthreads();
    {   // scale, skip connection, ReLU: 16 channels x 64 squares = one float4 per thread
        const int c = cg * 16 + (t >> 4), q = t & 15;
        const float4 v = reinterpret_cast<const float4 *>(xb + (size_t)c * 64)[q];
        float4 r = reinterpret_cast<float4 *>(rb + (size_t)c * 64)[q];
        const float bc = bias[c], g = gate16[t >> 4];
        r.x = (v.x + bc) * g + r.x; r.y = (v.y + bc) * g + r.y; r.z = (v.z + bc) * g + r.z; r.w = (v.w + bc) * g + r.w;
        r.x = r.x > 0.0f ? r.x : 0.0f; r.y = r.y > 0.0f ? r.y : 0.0f; r.z = r.z > 0.0f ? r.z : 0.0f; r.w = r.w > 0.0f ? r.w : 0.0f;
        reinterpret_cast<float4 *>(rb + (size_t)c * 64)[q] = r;
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
