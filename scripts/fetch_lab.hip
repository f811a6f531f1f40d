// scripts/fetch_lab.hip -- LAB: how fast can ONE compute unit stream 147 KB of weights (the per-layer fetch of bo_k_tower_b1)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/fetch_lab.hip -o /tmp/fetch_lab && /tmp/fetch_lab
// Every workgroup streams `layers` x 147 456 bytes from its own region of a 400 MB buffer (cold: HBM / Infinity Cache), `waves`
// waves per workgroup each loading bytes / waves as 16-byte-per-lane loads, all of a layer's loads issued before the first is used.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NL>  // loads per lane per layer
__global__ void __launch_bounds__(1024) k_fetch(const f4 *w, float *out, int layers, size_t wg_stride4, size_t layer_stride4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 *p = w + (size_t)blockIdx.x * wg_stride4 + (size_t)wave * NL * 64 + lane;
    f4 acc = {0, 0, 0, 0};
    for (int l = 0; l < layers; l++) {
        f4 a[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) a[i] = p[(size_t)l * layer_stride4 + (size_t)i * 64];
#pragma unroll
        for (int i = 0; i < NL; i++) acc += a[i];
        __syncthreads();
    }
    if (acc[0] == 123.456f) out[blockIdx.x] = acc[1] + acc[2] + acc[3];
}

int main() {
    const size_t layer_bytes = 147456, layers = 40;
    const int wgs = 64;
    const size_t wg_bytes = layer_bytes * layers;
    f4 *w; float *out;
    CK(hipMalloc(&w, wg_bytes * wgs)); CK(hipMemset(w, 0, wg_bytes * wgs)); CK(hipMalloc(&out, 4096));
    char *flush; CK(hipMalloc(&flush, 600u << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves : {4, 8, 16}) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(flush, rep, 600u << 20));  // evict the Infinity Cache
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const size_t ws4 = wg_bytes / 16, ls4 = layer_bytes / 16;
            if (waves == 4) hipLaunchKernelGGL((k_fetch<36>), dim3(wgs), dim3(256), 0, 0, w, out, (int)layers, ws4, ls4);
            else if (waves == 8) hipLaunchKernelGGL((k_fetch<18>), dim3(wgs), dim3(512), 0, 0, w, out, (int)layers, ws4, ls4);
            else hipLaunchKernelGGL((k_fetch<9>), dim3(wgs), dim3(1024), 0, 0, w, out, (int)layers, ws4, ls4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%2d waves per workgroup, %d workgroups: %.1f us per layer, %.1f GB/s per CU, %.2f TB/s chip\n", waves, wgs, ms * 1e3 / layers,
                   layer_bytes / (ms * 1e-3 / layers) / 1e9, layer_bytes * wgs / (ms * 1e-3 / layers) / 1e12);
        }
    }
    return 0;
}
