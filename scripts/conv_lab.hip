// scripts/conv_lab.hip -- timing lab for csrc/bo_conv.h (build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/conv_lab.hip -o /tmp/conv_lab && /tmp/conv_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../betaone_amd/csrc/bo_conv.h"

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

int main() {
    const int BMAX = 2048, C = 256;
    float *x, *y, *res, *bias; bo_f32x4 *w;
    CK(hipMalloc(&x, (size_t)BMAX * C * 64 * 4)); CK(hipMalloc(&y, (size_t)BMAX * C * 64 * 4)); CK(hipMalloc(&res, (size_t)BMAX * C * 64 * 4));
    CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&w, (size_t)9 * C * C * 4));
    CK(hipMemset(x, 0, (size_t)BMAX * C * 64 * 4)); CK(hipMemset(res, 0, (size_t)BMAX * C * 64 * 4)); CK(hipMemset(bias, 0, C * 4));
    CK(hipMemset(w, 0, (size_t)9 * C * C * 4));
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
    CK(hipDeviceSynchronize());
    return 0;
}
