// scripts/mfma_shape_lab.hip -- LAB (round 5): does the tile shape of the fp16 MFMA change the clock the chip holds under load?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_shape_lab.hip -o /tmp/mfma_shape_lab && /tmp/mfma_shape_lab
// bo_k_tower_s multiplies with v_mfma_f32_32x32x16_f16; with four cohorts' towers in flight a launch needs the same ~470 k shader
// cycles as alone but takes 218 us instead of 186: the clock.  MI355X_MICROARCH.md (DVFS give-back, item 7) reports bare bf16 loops
// of 16x16x32 tiles on RANDOM data at 1.12-1.15 x the FLOP/s of 32x32x16 loops at equal cycles per FLOP.  Round 3's lab measured both
// shapes on low-entropy operands (1.91 against 1.90 PFLOP/s).  Here: random fp16 operands re-read from LDS by ds_read_b128 every step
// (as the tower does), one wave per SIMD, same flops per wave, on 64 and on 256 compute units, with the in-kernel clock
// (s_memtime / s_memrealtime) of every run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// SHAPE 0: a 32 x 64 output tile per wave as 2 accumulators of 32x32, K-step 16: 6 MFMAs (3 per (hi, lo) product) of 32 cycles
// SHAPE 1: the same tile as 8 accumulators of 16x16, K-step 32: 24 MFMAs of 16 cycles.  Same operand bytes per flop.
template <int SHAPE>
__global__ void __launch_bounds__(256) k_loop(const h8 *src, float *out, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) h8 L[4096];  // 64 KiB of random halves
    for (int i = threadIdx.x; i < 4096; i += 256) L[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, c0 = 0;
    if (threadIdx.x == 0) { t0 = wall_clock64(); c0 = __builtin_readcyclecounter(); }
    float s = 0;
    if (SHAPE == 0) {
        f16v acc[2];
        for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int base = ((it * 8 + u) * 6 * 64 + wave * 1024) & 4095;
                const h8 ah = L[(base + lane) & 4095], al = L[(base + 64 + lane) & 4095];
                const h8 b0h = L[(base + 128 + lane) & 4095], b0l = L[(base + 192 + lane) & 4095], b1h = L[(base + 256 + lane) & 4095], b1l = L[(base + 320 + lane) & 4095];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc[1], 0, 0, 0);
            }
            // keep the sums finite: random products of magnitude ~1 would overflow nothing in fp32, but damp anyway
            if ((it & 63) == 63) for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] *= 0.5f;
        }
        for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    } else {
        f4v acc[8];
        for (int i = 0; i < 8; i++) acc[i] = f4v{0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {  // one K-step of 32 = two of the other shape's: 12 operand reads, 24 MFMAs
                const int base = ((it * 4 + u) * 12 * 64 + wave * 1024) & 4095;
                h8 a[2][2], b[4][2];
#pragma unroll
                for (int i = 0; i < 2; i++) for (int hl = 0; hl < 2; hl++) a[i][hl] = L[(base + (i * 2 + hl) * 64 + lane) & 4095];
#pragma unroll
                for (int i = 0; i < 4; i++) for (int hl = 0; hl < 2; hl++) b[i][hl] = L[(base + 256 + (i * 2 + hl) * 64 + lane) & 4095];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], acc[i * 4 + j], 0, 0, 0);
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], acc[i * 4 + j], 0, 0, 0);
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], acc[i * 4 + j], 0, 0, 0);
                    }
            }
            if ((it & 63) == 63) for (int i = 0; i < 8; i++) acc[i] *= 0.5f;
        }
        for (int i = 0; i < 8; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = wall_clock64() - t0; clk[2 * blockIdx.x + 1] = __builtin_readcyclecounter() - c0; }
}

int main() {
    h8 *src; float *out; unsigned long long *clk;
    CK(hipMalloc(&src, 65536)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int random_data = 0; random_data < 2; random_data++) {
        std::vector<_Float16> h(32768);
        unsigned r = 99u;
        for (auto &v : h) { r = r * 1664525u + 1013904223u; v = random_data ? (_Float16)((((r >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.5f) : (_Float16)0.0f; }
        CK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
        for (int wgs : {64, 256})
            for (int shape = 0; shape < 2; shape++) {
                const int iters = 40000;  // 40000 x 48 MFMA-equivalents of 32 cycles = 61 M cycles ~ 30 ms: long enough for the clock to settle
                float best = 1e9f; double ghz = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    if (shape == 0) hipLaunchKernelGGL((k_loop<0>), dim3(wgs), dim3(256), 0, 0, src, out, iters, clk);
                    else hipLaunchKernelGGL((k_loop<1>), dim3(wgs), dim3(256), 0, 0, src, out, iters, clk);  // (4 K-steps of 32 = the other shape's 8 of 16)
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) {
                        best = ms;
                        std::vector<unsigned long long> hc(2 * wgs); CK(hipMemcpy(hc.data(), clk, 16 * wgs, hipMemcpyDeviceToHost));
                        double a = 0, b = 0; for (int i = 0; i < wgs; i++) { a += hc[2 * i]; b += hc[2 * i + 1]; }
                        ghz = b / (a * 10.0);
                    }
                }
                const double flops = (double)iters * 48 * 32768.0 * 4 * wgs;  // per wave 48 MFMAs of 32x32x16 (or their equivalent) per iteration
                const double cyc_per_32 = best * 1e-3 * ghz * 1e9 / ((double)iters * 48);
                printf("%s operands, %3d CUs, %s: %7.2f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz  %.1f cycles per 32x32x16-equivalent\n",
                       random_data ? "random" : "zero  ", wgs, shape == 0 ? "32x32x16" : "16x16x32", best, flops / (best * 1e-3) / 1e12, ghz, cyc_per_32);
            }
    }
    return 0;
}
