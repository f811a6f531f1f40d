// betaone_amd/csrc/bo_select_wide.h -- PUCT selection over WIDE trees: the HBM-roofline form of the
// select kernel (SURVEY.md section 8d "select-kernel roofline workload"; BASELINE.json north_star:
// ">= 40 % of HBM roofline on the PUCT-select kernel").
//
// Same arithmetic as MCTSNode.select_child (/root/reference/mcts.py:72-118, regime R3) and the same
// first-maximum tie-break as bo_tree.h's select_leaf, but laid out for bandwidth instead of for one
// small tree per game:
//   * children of one node form one 512-byte, 512-byte-aligned CHILD BLOCK of 32 16-byte records
//       { int32 n; float q; float prior; int32 child_block }
//     so a level of the descent is ONE global_load_dwordx4 per lane = one coalesced 512-B request per
//     half-wave (12 B/child of statistics + the 4-B link, which shares the cache lines anyway);
//   * 32 children = half a wavefront; every half-wave keeps UT (=4) independent trees in flight and
//     issues all their loads before using any (the descent is a chain of dependent reads, so
//     bandwidth comes from the number of trees in flight: 8 x 512 B per wave, 32 waves per CU);
//   * a block is read once per launch: non-temporal loads, nothing is worth keeping in L2;
//   * blockIdx is consumed in the dispatcher's round-robin XCD order, so consecutive tree ids land
//     on different XCDs and their L2s share nothing that matters (trees are private).
// Algorithmic bytes per level = 12*32 + 8 = 392 B; the kernel returns the
// number of levels it actually descended so bench.py prices measured, not assumed, traffic.
#pragma once
#include "bo_wave.h"

#define BO_WIDE_C 32
struct WideChild {
    int n;            // visit count
    float q;          // mean value
    float prior;      // prior probability
    int child_block;  // index of this child's own child block, -1 = not expanded
};
struct alignas(512) WideBlock {
    WideChild child[BO_WIDE_C];
};


#if !defined(BO_WAVE_EMU)
typedef int wide_i4 __attribute__((ext_vector_type(4)));  // one WideChild as a 16-byte vector
template <bool NT> __device__ __forceinline__ wide_i4 wide_ld(const wide_i4 *v) {
    return NT ? __builtin_nontemporal_load(v) : *v;
}

template <int UT, bool NT> __device__ __forceinline__ void
select_wide_body(const wide_i4 *__restrict__ recs /* WideBlock[] viewed as 16-byte records */, const int *__restrict__ root_block, const int *__restrict__ root_n,
                 const float *__restrict__ sqrt_lut, int n_trees, int max_depth, float cpuct, int *__restrict__ out_leaf,
                 int *__restrict__ out_levels) {
    const int half = (int)(threadIdx.x >> 5);  // 8 half-waves per workgroup, 32 lanes = 32 children
    const int c = (int)(threadIdx.x & 31);
    const int stride = (int)gridDim.x * 8 * UT;
    for (int base = ((int)blockIdx.x * 8 + half) * UT; __ballot(base < n_trees) != 0; base += stride) {
        int blk[UT], pv[UT], pvn[UT], lev[UT], leaf[UT];
#pragma unroll
        for (int u = 0; u < UT; u++) {
            const int t = base + u;
            const bool ok = t < n_trees;
            blk[u] = ok ? root_block[t] : -1;
            pv[u] = pvn[u] = ok ? root_n[t] : 0;
            lev[u] = 0;
            leaf[u] = -1;
        }
        bool any = true;
        while (__ballot(any) != 0) {
            int n[UT], cb[UT];
            float q[UT], p[UT], sp[UT];
            // issue every tree's 512-B child block (4 coalesced 128-B lines) before touching any of them
#pragma unroll
            for (int u = 0; u < UT; u++) {
                const bool act = blk[u] >= 0;
                const wide_i4 r = wide_ld<NT>(recs + (size_t)(act ? blk[u] : 0) * BO_WIDE_C + c);
                const int r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                n[u] = act ? r0 : 0;
                q[u] = __int_as_float(r1);
                p[u] = __int_as_float(r2);
                cb[u] = act ? r3 : -1;
                sp[u] = sqrt_lut[pv[u]];
            }
            any = false;
#pragma unroll
            for (int u = 0; u < UT; u++) {
                const bool act = blk[u] >= 0;
                const float t1 = cpuct * p[u];
                const float t2 = t1 * sp[u];
                float score = n[u] > 0 ? q[u] + t2 / (float)(1 + n[u]) : 0.0f + t2;
                if (!(score == score)) score = -__builtin_inff();
                int bi = c, bn = n[u];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    const float os = __shfl_xor(score, m, 64);
                    const int oi = __shfl_xor(bi, m, 64);
                    const int on = __shfl_xor(bn, m, 64);
                    if (os > score || (os == score && oi < bi)) { score = os; bi = oi; bn = on; }
                }
                const int next = __shfl(cb[u], (int)(threadIdx.x & 32) + bi, 64);
                if (act) {
                    leaf[u] = blk[u] * BO_WIDE_C + bi;
                    lev[u] += 1;
                    // mcts.py:89: a non-root node scans its children with ITS PARENT's visit count
                    pv[u] = pvn[u];
                    pvn[u] = bn;
                    blk[u] = lev[u] < max_depth ? next : -1;
                }
                any = any || blk[u] >= 0;
            }
        }
#pragma unroll
        for (int u = 0; u < UT; u++)
            if (c == 0 && base + u < n_trees) { out_leaf[base + u] = leaf[u]; out_levels[base + u] = lev[u]; }
    }
}
#define BO_WIDE_KERNEL(NAME, U, NT)                                                                                     \
    extern "C" __global__ void __launch_bounds__(256)                                                                   \
    NAME(const WideBlock *__restrict__ blocks, const int *__restrict__ root_block, const int *__restrict__ root_n,       \
         const float *__restrict__ sqrt_lut, int n_trees, int max_depth, float cpuct, int *__restrict__ out_leaf,        \
         int *__restrict__ out_levels) {                                                                                \
        select_wide_body<U, NT>(reinterpret_cast<const wide_i4 *>(blocks), root_block, root_n, sqrt_lut, n_trees, max_depth, cpuct, out_leaf, out_levels);   \
    }
BO_WIDE_KERNEL(bo_k_select_wide, 4, true)  // measured: 2 and 8 trees per half-wave and plain (cached) loads are all slower
#endif
