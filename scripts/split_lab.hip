// scripts/split_lab.hip -- timing lab for csrc/bo_tower_s.h (build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/split_lab.hip -o /tmp/split_lab && /tmp/split_lab
// Zero weights and planes; what is timed is the structure: the full kernel, then with the weight loads, the B operand
// reads, the epilogue or all three removed, with every layer reading the same (L2-resident) weights, and a bare MFMA loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../betaone_amd/csrc/bo_tower_s.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_mfma_peak_h(float *out, int iters) {
    bo_f32x16 acc[2];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
    bo_h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mfma_peak_h16(float *out, int iters) {
    bo_f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = bo_f32x4{0, 0, 0, 0};
    bo_h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int C, int MT>
static int lab(int blocks, int B, bool random_data = false) {
    const int NL = 2 * blocks + 1;
    const size_t per = (size_t)(9 * C / 16) * (C / 32) * 2 * 64;  // bo_h8 per layer
    bo_h8 *tw; float *tp, *x, *oa, *ob; bo_tower_layer *tl;
    CK(hipMalloc(&tw, per * NL * 16 + (1 << 20))); CK(hipMemset(tw, 0, per * NL * 16 + (1 << 20)));
    CK(hipMalloc(&tp, 64 * 1024 * 4)); CK(hipMemset(tp, 0, 64 * 1024 * 4));
    CK(hipMalloc(&x, (size_t)B * 120 * 64 * 4)); CK(hipMemset(x, 0, (size_t)B * 120 * 64 * 4));
    if (random_data) {  // weights: hi halves of magnitude ~2^-6..2^-3 (after the 1/s of the epilogue: s = 2^-12), lo halves 2^-11 of that; planes 0 / 1
        printf("-- random weights and planes (activations stay O(1)) --\n");
        std::vector<_Float16> hw(per * NL * 8);
        unsigned r = 12345u;
        for (size_t i = 0; i < hw.size(); i++) {
            r = r * 1664525u + 1013904223u;
            const float u = ((r >> 8) & 0xffff) / 65536.0f - 0.5f;
            const bool lo = (i / 512) & 1;  // [..][hi | lo][64 lanes][8]
            hw[i] = (_Float16)(lo ? u * 0.25f : u * 512.0f);
        }
        CK(hipMemcpy(tw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)B * 120 * 64);
        for (size_t i = 0; i < hx.size(); i++) { r = r * 1664525u + 1013904223u; hx[i] = (r >> 20) & 1 ? 1.0f : 0.0f; }
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hp(64 * 1024, 0.0f);
        for (int l = 0; l < NL; l++) hp[l * (C + 4) + C] = 1.0f / (512.0f * 0.29f * 48.0f);  // keeps the activations' scale from layer to layer (roughly)
        CK(hipMemcpy(tp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&tl, NL * sizeof(bo_tower_layer)));
    CK(hipMalloc(&oa, (size_t)B * 128 * 4)); CK(hipMalloc(&ob, (size_t)B * 2048 * 4));
    std::vector<bo_tower_layer> L(NL);
    for (int l = 0; l < NL; l++) L[l] = {(int)(l * per), l == 0 ? 72 : 9 * C / 16, l * (C + 4), l == 0 ? 0 : (l % 2 ? 1 : 2), 0, 0, 0, l == NL - 1};

    CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
    bo_tower_head_s hh; hh.channels = 34; hh.split = 2; hh.w_off8 = (int)(NL * per); hh.b_off = 60000; hh.out_a = oa; hh.out_b = ob;
    const char *names[11] = {"full", "no weight loads", "no B reads", "no epilogue", "MFMA loop only", "full, every layer the same weights",
                             "B reads into a dead register set", "B operands read two K-steps ahead", "B three K-steps ahead",
                             "B three ahead, weights 12 ahead", "B two ahead, weights 12 ahead"};
    for (int variant = 0; variant < 11; variant++) {
        if (random_data && variant != 0 && variant != 4 && variant != 10) continue;
        if (variant == 5) {
            for (int l = 1; l < NL; l++) L[l].w_off4 = (int)(1 * per);
            CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
        }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto go = [&]() {
            const dim3 g(B < 256 ? B : 256), t(256);
            if (variant == 1) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 1>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 2) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 2>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 3) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 4>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 4) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 5>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 6) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 3>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 7) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 0, 2>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 8) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 0, 3>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 9) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 0, 3, 12>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else if (variant == 10) hipLaunchKernelGGL((bo_k_tower_s<C, MT, 0, 2, 12>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
            else hipLaunchKernelGGL((bo_k_tower_s<C, MT, 0>), g, t, 0, 0, x, tw, tp, tl, NL, nullptr, B, hh);
        };
        for (int i = 0; i < 3; i++) go();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 10; i++) go();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("split tower %d+0x%d (%s) B=%d: %.1f us = %.2f us/layer/256 boards\n", blocks, C, names[variant], B, ms * 1000 / 10,
               ms * 1000 / 10 / NL / ((B + 255) / 256));
    }
    return 0;
}

// fp16 tower, 128 filters, two boards per workgroup: prefetch depths (BD, AR) at a fast-mode batch
template <int BD, int AR>
static int lab_h(int blocks, int B) {
    constexpr int C = 128;
    const int NL = 2 * blocks + 1;
    const size_t per = (size_t)(9 * C / 16) * (C / 32) * 64;  // bo_h8 per layer
    bo_h8 *tw; float *tp, *x; bo_tower_layer *tl; _Float16 *oa, *ob;
    CK(hipMalloc(&tw, per * NL * 16 + (1 << 20))); CK(hipMemset(tw, 0, per * NL * 16 + (1 << 20)));
    CK(hipMalloc(&tp, 64 * 1024 * 4)); CK(hipMemset(tp, 0, 64 * 1024 * 4));
    CK(hipMalloc(&x, (size_t)B * 120 * 64 * 4)); CK(hipMemset(x, 0, (size_t)B * 120 * 64 * 4));
    CK(hipMalloc(&tl, NL * sizeof(bo_tower_layer)));
    CK(hipMalloc(&oa, (size_t)B * 128 * 2)); CK(hipMalloc(&ob, (size_t)B * 2048 * 2));
    std::vector<bo_tower_layer> L(NL);
    for (int l = 0; l < NL; l++) L[l] = {(int)(l * per), 72, l * (C + 4), l == 0 ? 0 : (l % 2 ? 1 : 2), 0, 0, 0, l == NL - 1};
    CK(hipMemcpy(tl, L.data(), NL * sizeof(bo_tower_layer), hipMemcpyHostToDevice));
    bo_tower_head_h hh; hh.channels = 34; hh.split = 2; hh.w_off8 = (int)(NL * per); hh.b_off = 60000; hh.out_a = oa; hh.out_b = ob;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto go = [&]() { hipLaunchKernelGGL((bo_k_tower_h<C, 1, 0, BD, AR>), dim3(256), dim3(256), 0, 0, x, tw, tp, tl, NL, B, hh); };
    for (int i = 0; i < 2; i++) go();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 5; i++) go();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("fp16 tower %d+0x128, B=%d, B operands %d steps ahead, weights %d ahead: %.1f us = %.2f us/layer/512 boards\n", blocks, B, BD, AR,
           ms * 1000 / 5, ms * 1000 / 5 / NL / (B / 512));
    (void)hipFree(tw); (void)hipFree(tp); (void)hipFree(x); (void)hipFree(tl); (void)hipFree(oa); (void)hipFree(ob);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1) {  // fp16 tower only
        if (lab_h<1, 8>(10, 4096) || lab_h<2, 8>(10, 4096) || lab_h<3, 8>(10, 4096) || lab_h<2, 12>(10, 4096) || lab_h<1, 12>(10, 4096)) return 1;
        if (lab_h<1, 8>(10, 512) || lab_h<2, 8>(10, 512) || lab_h<2, 12>(10, 512)) return 1;
        return 0;
    }
    {
        float *out; CK(hipMalloc(&out, 256 * 256 * 4));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2000;
        hipLaunchKernelGGL(k_mfma_peak_h, dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma_peak_h, dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 48;  // MFMAs per wave
        printf("bare v_mfma_f32_32x32x16_f16 chains, one wave per SIMD on 256 CUs: %.1f ns per MFMA = %.1f TFLOP/s\n", ms * 1e6 / n,
               n * 32768.0 * 1024 / (ms * 1e-3) / 1e12);
    }
    {
        float *out; CK(hipMalloc(&out, 256 * 256 * 4));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2000;
        hipLaunchKernelGGL(k_mfma_peak_h16, dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma_peak_h16, dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 96;  // MFMAs per wave
        printf("bare v_mfma_f32_16x16x32_f16 chains (8 accumulators, 3 dependent MFMAs each), one wave per SIMD on 256 CUs: %.1f ns per MFMA = %.1f TFLOP/s\n",
               ms * 1e6 / n, n * 16384.0 * 1024 / (ms * 1e-3) / 1e12);
    }
    if (lab<128, 1>(10, 256)) return 1;
    if (lab<128, 1>(10, 256, true)) return 1;
    if (lab<256, 2>(20, 256)) return 1;
    CK(hipDeviceSynchronize());
    return 0;
}
