// scripts/weight_stream_lab.hip -- LAB (round 5): what a Winograd form of the split-precision tower could cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/weight_stream_lab.hip -o /tmp/weight_stream_lab && /tmp/weight_stream_lab
// bo_k_tower_s keeps one board per workgroup, so EVERY compute unit streams the whole layer's weights through its own port to the
// XCD's L2: 589 824 B per 128-filter layer in the direct form ((hi, lo) fp16 pairs, 9 taps), 1 048 576 B in a Winograd F(2x2,3x3)
// form (16 transformed taps).  This lab times that stream by itself: W workgroups (one per CU, 4 waves) all read the SAME `layers`
// x `bytes` region (so all but the first reader of a line hit L2, as in the tower), 16 bytes per lane per load, RING loads per lane
// in flight, nothing else in the kernel.  If a 1 MB layer takes longer than the 8.9 us a direct layer takes today including its
// 1 728 MFMAs, fewer MFMAs cannot pay for the bigger stream at one board per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));

// every wave streams its quarter of each layer: loads [wave][i][lane] of 16 B, RING in flight, xor-folded so nothing is dropped
template <int RING>
__global__ void __launch_bounds__(256) k_stream(const i4 *w, int *out, int layers, int per_wave16, unsigned long long *clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<i4 *>(w), 0, 0x7fffffff, 0x00020000);
    i4 ring[RING], acc = {0, 0, 0, 0};
    const int total = layers * per_wave16;  // loads per lane over the whole run: layer-major, [layer][wave][i][lane]
    auto addr = [&](int k) { const int l = k / per_wave16, i = k - l * per_wave16; return (((l * 4 + wave) * per_wave16 + i) * 64 + lane) * 16; };
    unsigned long long t0 = 0, c0 = 0;
    if (threadIdx.x == 0) { t0 = wall_clock64(); c0 = __builtin_readcyclecounter(); }
#pragma unroll
    for (int j = 0; j < RING; j++) ring[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, addr(j), 0, 0);
    for (int k0 = 0; k0 < total; k0 += RING) {
#pragma unroll
        for (int j = 0; j < RING; j++) {
            acc ^= ring[j];
            const int kn = k0 + j + RING;
            ring[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, addr(kn < total ? kn : k0 + j), 0, 0);
        }
    }
    if (acc[0] == 0x12345678) out[blockIdx.x] = acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && clk) { clk[2 * blockIdx.x] = wall_clock64() - t0; clk[2 * blockIdx.x + 1] = __builtin_readcyclecounter() - c0; }
}

int main() {
    const int layers = 84;  // four towers' worth, so the stream does not fit one L2 (4 MiB): lines are re-fetched from the Infinity Cache as in the tower
    int *out; unsigned long long *clk; CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 256 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t bytes : {(size_t)589824, (size_t)1048576}) {
        i4 *w; CK(hipMalloc(&w, bytes * layers + 65536));
        std::vector<unsigned> h(bytes * layers / 4);
        unsigned r = 7u;
        for (auto &v : h) { r = r * 1664525u + 1013904223u; v = r; }
        CK(hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const int per_wave16 = (int)(bytes / 4 / 1024);  // 1 KiB wave-instructions per wave per layer
        for (int wgs : {64, 256})
            for (int ring : {12, 24, 48}) {
                float best = 1e9f; double ghz = 0;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipEventRecord(e0));
                    if (ring == 12) hipLaunchKernelGGL((k_stream<12>), dim3(wgs), dim3(256), 0, 0, w, out, layers, per_wave16, clk);
                    else if (ring == 24) hipLaunchKernelGGL((k_stream<24>), dim3(wgs), dim3(256), 0, 0, w, out, layers, per_wave16, clk);
                    else hipLaunchKernelGGL((k_stream<48>), dim3(wgs), dim3(256), 0, 0, w, out, layers, per_wave16, clk);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) {
                        best = ms;
                        unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
                        ghz = (double)hc[1] / ((double)hc[0] * 10.0);  // shader cycles per 10 ns tick
                    }
                }
                const double us_layer = best * 1e3 / layers;
                printf("%7zu B per layer, %3d workgroups (one per CU), %2d x 1 KiB in flight per wave: %6.2f us per layer = %5.1f GB/s per CU = %4.1f B/clk at %.2f GHz\n",
                       bytes, wgs, ring, us_layer, bytes / us_layer / 1e3, bytes / (us_layer * 1e3 * ghz), ghz);
            }
        CK(hipFree(w));
    }
    printf("reference: bo_k_tower_s today, lone 64-board launch: 186 us / 21 layers = 8.9 us per direct layer INCLUDING its 1 728 MFMAs per board\n");
    return 0;
}
