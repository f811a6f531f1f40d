// scripts/conv_lab.hip -- timing lab for csrc/bo_conv.h (build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/conv_lab.hip -o /tmp/conv_lab && /tmp/conv_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../betaone_amd/csrc/bo_conv.h"
#include "../betaone_amd/csrc/bo_tower.h"
#include "../betaone_amd/csrc/bo_tower_wg.h"
#include "../betaone_amd/csrc/bo_tower_h.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CI, int CO, int LAB>
static float run(int B, const float *x, const bo_f32x4 *w, const float *bias, const float *res, float *y, int mode, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((bo_k_conv3x3<CI, CO, LAB>), dim3(B), dim3(CO * 2), 0, 0, x, w, bias, res, y, mode);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((bo_k_conv3x3<CI, CO, LAB>), dim3(B), dim3(CO * 2), 0, 0, x, w, bias, res, y, mode);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}

// sustained MFMA ceiling: 256 workgroups x 4 waves, nothing but v_mfma_f32_32x32x2_f32 on NACC accumulators
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_peak(float *out, int iters, float a, float b) {
    bo_f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = bo_f32x16{0};
    float av = a + threadIdx.x, bw = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw + i, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma16_peak(float *out, int iters, float a, float b) {
    bo_f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = bo_f32x4{0, 0, 0, 0};
    float av = a + threadIdx.x, bw = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw + i, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void peak16(float *out) {
    const int iters = 1024 * 20 / NACC;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_mfma16_peak<NACC>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_mfma16_peak<NACC>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / 10, n_mfma = (double)iters * NACC;
    printf("pure MFMA 16x16x4 f32, %d accumulators: %.1f us for %.0f MFMAs/wave = %.1f ns/MFMA -> %.1f TFLOP/s\n", NACC, us, n_mfma,
           us * 1000 / n_mfma, n_mfma * 2048 * 4 * 256 / us / 1e6);
}

template <int NACC, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_mfma_h_peak(float *out, int iters, float a, float b) {
    bo_f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
    bo_h8 av, bw[NACC];
    for (int e = 0; e < 8; e++) { av[e] = (_Float16)(a + threadIdx.x + e); for (int i = 0; i < NACC; i++) bw[i][e] = (_Float16)(b + i + e); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bw[i], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}
template <int NACC, int WAVES>
static void peak_h(float *out) {
    const int per_wave = 1152 * 40 * 4 / WAVES;  // MFMAs per wave: 40 layers of the 256-filter tower per SIMD
    const int iters = per_wave / (4 * NACC);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_mfma_h_peak<NACC, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_mfma_h_peak<NACC, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / 10, n_mfma = (double)iters * 4 * NACC;
    printf("pure MFMA 32x32x16 f16, %d accumulators, %d waves/CU: %.1f us for %.0f MFMAs/wave = %.1f ns/MFMA/SIMD -> %.0f TFLOP/s (%.2f GHz at 32 cycles)\n", NACC,
           WAVES, us, n_mfma, us * 1000 / (n_mfma * WAVES / 4), n_mfma * 32768.0 * WAVES * 256 / us / 1e6, 32.0 / (us * 1000 / (n_mfma * WAVES / 4)));
}

template <int NACC>
static void peak(float *out, int grid) {
    const int iters = 1152 / (8 * NACC) * 20;  // 20 conv layers' worth of MFMAs per wave
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_mfma_peak<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_mfma_peak<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / 10, n_mfma = (double)iters * 8 * NACC;
    printf("pure MFMA, %d accumulators, grid %d: %.1f us for %.0f MFMAs/wave = %.1f ns/MFMA -> %.1f TFLOP/s (%.2f GHz at 64 cycles)\n", NACC, grid, us,
           n_mfma, us * 1000 / n_mfma, n_mfma * 4096 * 4 * grid / us / 1e6, 64.0 / (us * 1000 / n_mfma));
}

int main() {
    const int BMAX = 2048, C = 256;
    float *x, *y, *res, *bias; bo_f32x4 *w;
    CK(hipMalloc(&x, (size_t)BMAX * C * 64 * 4)); CK(hipMalloc(&y, (size_t)BMAX * C * 64 * 4)); CK(hipMalloc(&res, (size_t)BMAX * C * 64 * 4));
    CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&w, (size_t)9 * C * C * 4));
    CK(hipMemset(x, 0, (size_t)BMAX * C * 64 * 4)); CK(hipMemset(res, 0, (size_t)BMAX * C * 64 * 4)); CK(hipMemset(bias, 0, C * 4));
    CK(hipMemset(w, 0, (size_t)9 * C * C * 4));
    peak_h<4, 4>(y); peak_h<4, 8>(y); peak_h<2, 8>(y);
    peak16<4>(y); peak16<8>(y); peak16<32>(y);
    peak<1>(y, 256); peak<2>(y, 256); peak<4>(y, 256); peak<2>(y, 32); peak<2>(y, 512);
    for (int B : {256, 512, 768, 1024, 2048}) {
        float t0 = run<128, 128, 0>(B, x, w, bias, res, y, 2, 50);
        float t1 = run<128, 128, 1>(B, x, w, bias, res, y, 2, 50);
        float t2 = run<128, 128, 2>(B, x, w, bias, res, y, 2, 50);
        float t3 = run<128, 128, 3>(B, x, w, bias, res, y, 2, 50);
        printf("128x128 B=%4d: full %.1f us (%.1f TFLOP/s)  no-loop %.1f  no-lds %.1f  no-weights %.1f\n", B, t0,
               (double)B * 128 * 128 * 9 * 64 * 2 / t0 / 1e6, t1, t2, t3);
    }
    for (int B : {256, 1024}) {
        float a = run<256, 256, 0>(B, x, w, bias, res, y, 2, 20);
        float c = run<64, 64, 0>(B, x, w, bias, res, y, 2, 50);
        float d = run<120, 128, 0>(B, x, w, bias, res, y, 1, 50);
        printf("B=%4d: 256x256 %.1f us (%.1f TF)  64x64 %.1f us (%.1f TF)  120->128 %.1f us\n", B, a, (double)B * 256 * 256 * 9 * 128 / a / 1e6, c,
               (double)B * 64 * 64 * 9 * 128 / c / 1e6, d);
    }
    // whole tower, 8 plain + 2 SE blocks x 128 filters: distinct weights per layer vs one shared (L2-hot) set
    {
        const int C = 128, NL = 21;
        const size_t per = (size_t)9 * 16 * C * 2;  // float4 per layer
        bo_f32x4 *tw; float *tp; bo_tower_layer *tl;
        CK(hipMalloc(&tw, per * NL * 16)); CK(hipMemset(tw, 0, per * NL * 16));
        CK(hipMalloc(&tp, 64 * 1024 * 4)); CK(hipMemset(tp, 0, 64 * 1024 * 4));
        CK(hipMalloc(&tl, NL * sizeof(bo_tower_layer)));
        for (int variant = 0; variant < 3; variant++) {
            std::vector<bo_tower_layer> L(NL);
            for (int l = 0; l < NL; l++) {
                const bool se = variant != 2 && l >= 17 && (l % 2 == 0);
                L[l] = {(int)((variant == 1 ? 0 : l) * per), 16, l * 128, l == 0 ? 0 : (l % 2 ? 1 : (se ? 3 : 2)), 4096, 8192, 8, l == NL - 1};
            }
            CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
            for (int B : {256, 512}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                for (int i = 0; i < 3; i++) hipLaunchKernelGGL((bo_k_tower<128>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, y, B);
                (void)hipEventRecord(e0, 0);
                for (int i = 0; i < 20; i++) hipLaunchKernelGGL((bo_k_tower<128>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, y, B);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                printf("tower variant %d (%s) B=%d: %.1f us = %.2f us/layer/256 boards\n", variant,
                       variant == 0 ? "distinct weights" : variant == 1 ? "shared weights" : "distinct, no SE", B, ms * 1000 / 20,
                       ms * 1000 / 20 / NL / (B / 256));
            }
        }
    }
    // Winograd tower: 16*128*128 floats per layer = 65536 float4
    {
        const int NL = 21; const size_t per = 65536;
        bo_f32x4 *tw; float *tp; bo_tower_layer *tl;
        CK(hipMalloc(&tw, per * NL * 16)); CK(hipMemset(tw, 0, per * NL * 16));
        CK(hipMalloc(&tp, 64 * 1024 * 4)); CK(hipMemset(tp, 0, 64 * 1024 * 4));
        CK(hipMalloc(&tl, NL * sizeof(bo_tower_layer)));
        for (int variant = 0; variant < 7; variant++) {
            std::vector<bo_tower_layer> L(NL);
            for (int l = 0; l < NL; l++) L[l] = {(int)((variant == 1 ? 0 : l) * per), 32, l * 128, l == 0 ? 0 : (l % 2 ? 1 : 2), 0, 0, 0, l == NL - 1};
            CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
            for (int B : {256, 512}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                auto go = [&]() {
                    if (variant == 2) hipLaunchKernelGGL((bo_k_tower_wg<128, 1>), dim3(256), dim3(512), 0, 0, x, tw, tp, tl, NL, y, B);
                    else if (variant == 3) hipLaunchKernelGGL((bo_k_tower_wg<128, 2>), dim3(256), dim3(512), 0, 0, x, tw, tp, tl, NL, y, B);
                    else if (variant == 4) hipLaunchKernelGGL((bo_k_tower_wg<128, 4>), dim3(256), dim3(512), 0, 0, x, tw, tp, tl, NL, y, B);
                    else if (variant == 5) hipLaunchKernelGGL((bo_k_tower_wg<128, 5>), dim3(256), dim3(512), 0, 0, x, tw, tp, tl, NL, y, B);
                    else if (variant == 6) hipLaunchKernelGGL((bo_k_tower_wg<128, 6>), dim3(256), dim3(512), 0, 0, x, tw, tp, tl, NL, y, B);
                    else hipLaunchKernelGGL((bo_k_tower_wg<128, 0>), dim3(256), dim3(512), 0, 0, x, tw, tp, tl, NL, y, B);
                };
                for (int i = 0; i < 3; i++) go();
                (void)hipEventRecord(e0, 0);
                for (int i = 0; i < 20; i++) go();
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                const char *names[7] = {"distinct weights", "shared weights", "no weight loads", "no input transform", "no epilogue", "no chunk barriers", "MFMA + B reads only"};
                printf("winograd tower (%s, no SE) B=%d: %.1f us = %.2f us/layer/256 boards\n", names[variant], B,
                       ms * 1000 / 20, ms * 1000 / 20 / NL / (B / 256));
            }
        }
    }
    // fp16 tower, 20 blocks x 256 filters (no SE), 512 boards = one pair per CU
    {
        const int NL = 41; const size_t per = (size_t)144 * 8 * 64;  // bo_h8 per layer
        bo_h8 *tw; float *tp; bo_tower_layer *tl; _Float16 *oa, *ob;
        CK(hipMalloc(&tw, per * NL * 16 + 65536)); CK(hipMemset(tw, 0, per * NL * 16 + 65536));
        CK(hipMalloc(&tp, 64 * 1024 * 4)); CK(hipMemset(tp, 0, 64 * 1024 * 4));
        CK(hipMalloc(&tl, NL * sizeof(bo_tower_layer)));
        CK(hipMalloc(&oa, 2048 * 128 * 2)); CK(hipMalloc(&ob, 2048 * 2048 * 2));
        std::vector<bo_tower_layer> L(NL);
        for (int l = 0; l < NL; l++) L[l] = {(int)(l * per), l == 0 ? 72 : 144, l * 256, l == 0 ? 0 : (l % 2 ? 1 : 2), 0, 0, 0, l == NL - 1};
        CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
        bo_tower_head_h hh; hh.channels = 34; hh.split = 2; hh.w_off8 = (int)(NL * per); hh.b_off = 0; hh.out_a = oa; hh.out_b = ob;
        for (int variant = 0; variant < 7; variant++) {
            if (variant == 6) {  // every layer reads layer 1's weights: they stay in L2 (is the weight stream latency or bandwidth?)
                for (int l = 1; l < NL; l++) L[l].w_off4 = (int)(1 * per);
                CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
            }
            for (int B : {512}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                auto go = [&]() {
                    if (variant == 1) hipLaunchKernelGGL((bo_k_tower_h<256, 2, 1>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh);
                    else if (variant == 2) hipLaunchKernelGGL((bo_k_tower_h<256, 2, 2>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh);
                    else if (variant == 3) hipLaunchKernelGGL((bo_k_tower_h<256, 2, 4>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh);
                    else if (variant == 4) hipLaunchKernelGGL((bo_k_tower_h<256, 2, 5>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh);
                    else if (variant == 5) hipLaunchKernelGGL((bo_k_tower_h<256, 2, 6>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh);
                    else hipLaunchKernelGGL((bo_k_tower_h<256, 2, 0>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh);
                };
                for (int i = 0; i < 3; i++) go();
                (void)hipEventRecord(e0, 0);
                for (int i = 0; i < 10; i++) go();
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                const char *names[7] = {"full", "no weight loads", "no B reads", "no epilogue", "MFMA loop only", "MFMA loop only, no barriers", "full, shared weights"};
                printf("fp16 tower 20x256 (%s) B=%d: %.1f us = %.2f us/layer/512 boards\n", names[variant], B, ms * 1000 / 10, ms * 1000 / 10 / NL / (B / 512));
            }
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
