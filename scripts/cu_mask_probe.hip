// scripts/cu_mask_probe.hip -- LAB: which compute units a stream made with hipExtStreamCreateWithCUMask really runs on (MI355X, one
// partition of 8 XCDs x 32 CUs).  Every workgroup records (XCC_ID, HW_ID) and holds its CU for ~30 us so that the dispatcher has to
// spread the grid.   hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe scripts/cu_mask_probe.hip && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void k_where(unsigned *out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // XCC_ID[3:0]
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 3000) __builtin_amdgcn_s_sleep(8);  // 100 MHz clock: 30 us
}
static void run(const char *name, const std::vector<uint32_t> &mask, unsigned *dev, int wgs) {
    hipStream_t st;
    hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s stream: %s\n", name, hipGetErrorString(e)); return; }
    hipMemsetAsync(dev, 0xff, wgs * 8, st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    hipLaunchKernelGGL(k_where, dim3(wgs), dim3(256), 0, st, dev);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(2 * wgs);
    hipMemcpy(h.data(), dev, wgs * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int i = 0; i < wgs; i++) {
        const unsigned hw = h[2 * i + 1];
        per_xcc[h[2 * i] & 15].insert((hw >> 8) & 0xff);  // cu_id[11:8], sh_id[12], se_id[15:13]
    }
    int total = 0;
    printf("%-28s %4d workgroups %7.1f us  CUs per XCC:", name, wgs, ms * 1e3);
    for (auto &kv : per_xcc) { printf(" x%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  = %d distinct\n", total);
    hipStreamDestroy(st);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    unsigned *dev; hipMalloc(&dev, 4096 * 8);
    const int W = (p.multiProcessorCount + 31) / 32;
    run("no mask", {}, dev, 1024);
    for (int K : {2, 4, 8})
        for (int c = 0; c < K; c += K - 1) {
            std::vector<uint32_t> lo(W, 0), il(W, 0);
            const int per = p.multiProcessorCount / K;
            for (int i = 0; i < p.multiProcessorCount; i++) {
                if (i / per == c) lo[i / 32] |= 1u << (i % 32);
                if (i % K == c) il[i / 32] |= 1u << (i % 32);
            }
            char nm[64];
            snprintf(nm, sizeof nm, "K=%d part %d contiguous", K, c); run(nm, lo, dev, 1024); run(nm, lo, dev, per);
            snprintf(nm, sizeof nm, "K=%d part %d interleaved", K, c); run(nm, il, dev, 1024); run(nm, il, dev, per);
        }
    // Round 5 ("whole-XCD masks, two XCDs = 64 CUs per cohort").  The driver deals the mask's bits over the XCCs (bit i -> XCC i % 8,
    // position i / 8 inside it) and the hardware deals a grid's workgroups round-robin over ALL eight XCCs whatever the mask says, so a
    // mask that leaves an XCC without CUs cannot be honoured: these show what happens instead.
    for (int nx : {1, 2, 4}) {  // every CU of XCCs 0 .. nx-1, none elsewhere
        std::vector<uint32_t> m(W, 0);
        for (int i = 0; i < p.multiProcessorCount; i++) if (i % 8 < nx) m[i / 32] |= 1u << (i % 32);
        char nm[64]; snprintf(nm, sizeof nm, "whole XCCs 0..%d only", nx - 1);
        run(nm, m, dev, 1024); run(nm, m, dev, 64);
    }
    {   // every CU of XCCs 0 and 1, ONE CU in each of the other six (so that no XCC is empty)
        std::vector<uint32_t> m(W, 0);
        for (int i = 0; i < p.multiProcessorCount; i++) if (i % 8 < 2 || i / 8 == 0) m[i / 32] |= 1u << (i % 32);
        run("XCC 0,1 whole + 1 CU in others", m, dev, 1024); run("XCC 0,1 whole + 1 CU in others", m, dev, 64);
    }
    // what an unmasked stream does with a grid smaller than the chip (64 one-board workgroups, as a 64-board cohort's tower launches)
    run("no mask", {}, dev, 64); run("no mask", {}, dev, 128); run("no mask", {}, dev, 256);
    return 0;
}
