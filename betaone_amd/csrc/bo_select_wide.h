// betaone_amd/csrc/bo_select_wide.h -- PUCT selection over WIDE trees: the HBM-roofline form of the
// select kernel (SURVEY.md section 8d "select-kernel roofline workload"; BASELINE.json north_star:
// ">= 40 % of HBM roofline on the PUCT-select kernel").
//
// Same arithmetic as MCTSNode.select_child (/root/reference/mcts.py:72-118, regime R3) and the same
// first-maximum tie-break as bo_tree.h's select_leaf, but laid out for bandwidth instead of for one
// small tree per game:
//   * children of one node form one 512-byte, 512-byte-aligned CHILD BLOCK
//       int32 n[32] | float q[32] | float prior[32] | int32 child_block[32]
//     so a level of the descent is one 384-B coalesced read (3 x 128-B lines) + 4 B of the 4th line;
//   * 32 children = half a wavefront: each wavefront descends TWO trees at once, a 256-thread
//     workgroup eight, and the grid-stride loop keeps every CU's memory queue full (the descent is
//     a chain of dependent reads, so bandwidth comes from the number of trees in flight);
//   * blockIdx is consumed in the dispatcher's round-robin XCD order, so consecutive tree ids land
//     on different XCDs and their L2s share nothing that matters (trees are private).
// Algorithmic bytes per level = 12*32 + 8 = 392 B; the kernel returns the
// number of levels it actually descended so bench.py prices measured, not assumed, traffic.
#pragma once
#include "bo_wave.h"

#define BO_WIDE_C 32
struct WideBlock {
    int n[BO_WIDE_C];
    float q[BO_WIDE_C];
    float prior[BO_WIDE_C];
    int child_block[BO_WIDE_C];  // index of the child's own child block, -1 = leaf
};

#if !defined(BO_WAVE_EMU)
extern "C" __global__ void __launch_bounds__(256)
bo_k_select_wide(const WideBlock *__restrict__ blocks, const int *__restrict__ root_block, const int *__restrict__ root_n,
                 const float *__restrict__ sqrt_lut, int n_trees, int max_depth, float cpuct, int *__restrict__ out_leaf,
                 int *__restrict__ out_levels) {
    const int half = (int)(threadIdx.x >> 5);          // 8 half-waves per workgroup
    const int c = (int)(threadIdx.x & 31);
    const int stride = (int)gridDim.x * 8;
    for (int t = (int)blockIdx.x * 8 + half; t < n_trees; t += stride) {
        int blk = root_block[t];
        int pv = root_n[t], pv_next = pv;
        int levels = 0, leaf_code = -1;
        while (blk >= 0 && levels < max_depth) {
            const WideBlock *B = blocks + blk;
            const int n = B->n[c];
            const float p = B->prior[c];
            const float qv = B->q[c];
            const float sp = sqrt_lut[pv];
            const float t1 = cpuct * p;
            const float t2 = t1 * sp;
            float score = n > 0 ? qv + t2 / (float)(1 + n) : 0.0f + t2;
            if (!(score == score)) score = -__builtin_inff();
            int bi = c, bn = n;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                const float os = __shfl_xor(score, m, 64);
                const int oi = __shfl_xor(bi, m, 64);
                const int on = __shfl_xor(bn, m, 64);
                if (os > score || (os == score && oi < bi)) { score = os; bi = oi; bn = on; }
            }
            leaf_code = blk * BO_WIDE_C + bi;
            blk = B->child_block[bi];
            // mcts.py:89: a non-root node scans its children with ITS PARENT's visit count
            pv = pv_next;
            pv_next = bn;
            levels++;
        }
        if (c == 0) { out_leaf[t] = leaf_code; out_levels[t] = levels; }
    }
}
#endif
