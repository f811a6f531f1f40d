// betaone_amd/csrc/bo_conv.h -- 3x3 convolution of the residual tower on the fp32 matrix cores (gfx950 only).
//
// The evaluate stage of the path (PolicyValueNet, /root/reference/network.py:48-118,167-198) spends ~90 % of its
// time in 3x3 convolutions over 8x8 boards.  Under PyTorch-ROCm they run in MIOpen's fp32 Winograd kernel (54 us
// for [256,128,8,8] * [128,128,3,3]) followed by a separate bias/ReLU/residual pass.  This kernel is a direct
// implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation) with the epilogue fused:
//
//   D[oc][pos] = sum_{tap, ic} W[oc][ic][tap] * X[ic][pos shifted by tap]        M = oc, N = pos (64), K = 9*Cin
//
//   * one workgroup per board; wave w owns output channels 32w..32w+31 and both 32-position halves of the board
//     (two independent 32x32 accumulators = 32 AGPRs; the A fragment is shared by the two MFMAs of a K-step);
//   * the whole input board lives in LDS as a zero-padded 10x10 image, channel-interleaved so that the B operands
//     of four consecutive K-steps are ONE ds_read_b128: X[t4][k][padded pos][e] holds channel 8*t4 + 2*e + k.
//     The 9 taps are 9 constant offsets of one base address; no bounds checks in the loop;
//   * weights are pre-packed on the host as Wp[tap][Cin/8][oc][k=2][e=4] (value W[oc][8*t4 + 2*e + k][tap]): the
//     A fragments of the same four K-steps are ONE coalesced global_load_dwordx4 per lane (L2-resident, every
//     workgroup reads the same 9*Cin*Cout*4 bytes); the 9 taps of the NEXT channel group are loaded into a second
//     register set while the 72 MFMAs of the current group issue, and the B operands are read one tap ahead;
//   * epilogue in registers: y = relu(acc + bias[oc] (+ residual)), or acc + bias for the SE blocks.
// MFMA time at 100 % issue: 9*Cin/2 K-steps x 2 MFMAs x 64 cycles = 73.7k cycles = 30.7 us at Cin = Cout = 128.
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>

typedef float bo_f32x16 __attribute__((ext_vector_type(16)));
typedef float bo_f32x4 __attribute__((ext_vector_type(4)));

// Issue order of one tap: the two LDS reads and the weight load each go into the 64-cycle shadow of a different MFMA
// (issued as one clump after the 8th MFMA they take longer than that shadow and the matrix pipe idles ~10 %).
#define BO_CONV_TAP_SCHEDULE()                                 \
    do {                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);     \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

enum { BO_CONV_RAW_BIAS = 0, BO_CONV_BIAS_RELU = 1, BO_CONV_BIAS_RES_RELU = 2 };

// LAB (scripts/conv_lab.hip only): 0 = the kernel; 1 = no main loop; 2 = main loop without LDS reads;
// 3 = main loop without weight loads.
template <int CIN, int COUT, int LAB = 0>
__global__ void __launch_bounds__(COUT * 2)
bo_k_conv3x3(const float *__restrict__ x, const bo_f32x4 *__restrict__ wp, const float *__restrict__ bias,
             const float *__restrict__ res, float *__restrict__ y, int mode) {
    constexpr int T4 = CIN / 8;       // channel groups of 8 = 4 K-steps of 2 channels
    constexpr int PITCH = 100;        // padded 10x10 board
    constexpr int NT = COUT * 2;      // threads: one wave per 32 output channels
    __shared__ bo_f32x4 X[T4 * 2 * PITCH];  // [t4][k][padded pos] x 4 channels e
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int k = lane >> 5, j = lane & 31;
    const int oc = wave * 32 + j;                       // A operand row of this lane
    const bo_f32x4 *xl = &X[k * PITCH + ((j >> 3) + 1) * 10 + (j & 7) + 1];   // B operand: first half; second half = +40
    bo_f32x16 acc0 = {0}, acc1 = {0};
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) bv[r] = bias[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * k];
    const unsigned wlane = (unsigned)oc * 2 + k;        // Wp float4 index: ((tap*T4 + t4)*COUT)*2 [uniform] + oc*2 + k [lane]

    // Weight fragments ping-pong between two register sets.  The load of tap t of the NEXT channel group is issued
    // together with the LDS reads of the next tap, in the shadow of the 8 MFMAs of the current tap (a block of 9 loads
    // with their address arithmetic between two groups drains the MFMA pipe for ~600 cycles per group); the
    // sched_barriers pin that placement (left alone, the scheduler folds the two sets into one and issues each load
    // right before its first use).  Addresses are uniform base + constant lane offset: no VALU work per load.
    bo_f32x4 fa[9], fb[9];
    bo_f32x4 bn0, bn1;                                  // B operands of the next tap
    auto load = [&](bo_f32x4(&a)[9], int t4) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
            a[tap] = (LAB == 3) ? bo_f32x4{1.0f, 0.5f, 0.25f, 2.0f} : (wp + (size_t)(tap * T4 + t4) * COUT * 2)[wlane];
    };
    auto read_b = [&](int t4, int tap) {
        const int off = (tap / 3 - 1) * 10 + (tap % 3 - 1);
        if (LAB == 2) { bn0 = bo_f32x4{1.0f, 2.0f, 3.0f, 4.0f}; bn1 = bn0; return; }
        bn0 = xl[t4 * 2 * PITCH + off];
        bn1 = xl[t4 * 2 * PITCH + off + 40];
    };
    // 72 MFMAs of channel group t4 with fragments a; prefetches group t4 + 1 into an (if it exists)
    auto compute = [&](const bo_f32x4(&a)[9], bo_f32x4(&an)[9], int t4) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const bo_f32x4 b0 = bn0, b1 = bn1;
            if (tap < 8) read_b(t4, tap + 1);
            else read_b(t4 + 1 < T4 ? t4 + 1 : t4, 0);
            if (t4 + 1 < T4) an[tap] = (LAB == 3) ? bo_f32x4{1.0f, 0.5f, 0.25f, 2.0f} : (wp + (size_t)(tap * T4 + t4 + 1) * COUT * 2)[wlane];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap][e], b0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap][e], b1[e], acc1, 0, 0, 0);
            }
            BO_CONV_TAP_SCHEDULE();
        }
    };
    if (LAB != 1) load(fa, 0);  // first weight fragments fly while the board is staged

    // ---- stage the board: issue the global loads, zero the halo image while they fly, then scatter the squares ----
    constexpr int NLD = (CIN * 16 + NT - 1) / NT;  // 16 float4 (4 squares each) per channel
    const bo_f32x4 *xb = reinterpret_cast<const bo_f32x4 *>(x + (size_t)b * CIN * 64);
    bo_f32x4 stage[NLD];
#pragma unroll
    for (int u = 0; u < NLD; u++) {
        const int i = tid + u * NT;
        stage[u] = (CIN * 16 % NT == 0 || i < CIN * 16) ? xb[i] : bo_f32x4{0, 0, 0, 0};
    }
    for (int i = tid; i < T4 * 2 * PITCH; i += NT) X[i] = bo_f32x4{0, 0, 0, 0};
    __syncthreads();
    {
        float *Xf = reinterpret_cast<float *>(X);
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int i = tid + u * NT;
            if (CIN * 16 % NT == 0 || i < CIN * 16) {
                const int ic = i >> 4, q = i & 15, r = q >> 1, c0 = (q & 1) * 4;
                float *dst = Xf + (((ic >> 3) * 2 + (ic & 1)) * PITCH + (r + 1) * 10 + c0 + 1) * 4 + ((ic & 7) >> 1);
                dst[0] = stage[u][0]; dst[4] = stage[u][1]; dst[8] = stage[u][2]; dst[12] = stage[u][3];
            }
        }
    }
    __syncthreads();

    if (LAB != 1) {
        read_b(0, 0);
        int t4 = 0;
        for (; t4 + 2 <= T4; t4 += 2) {
            compute(fa, fb, t4);
            compute(fb, fa, t4 + 1);
        }
        if (t4 < T4) compute(fa, fb, t4);
    }

    // ---- epilogue: D row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 ----
    float *yb = y + (size_t)b * COUT * 64;
    const float *rb = res + (size_t)b * COUT * 64;
    float r0[16], r1[16];
    if (mode == BO_CONV_BIAS_RES_RELU) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
            r0[r] = rb[o * 64 + j];
            r1[r] = rb[o * 64 + 32 + j];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int o = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
        float v0 = acc0[r] + bv[r], v1 = acc1[r] + bv[r];
        if (mode == BO_CONV_BIAS_RES_RELU) { v0 += r0[r]; v1 += r1[r]; }
        if (mode != BO_CONV_RAW_BIAS) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
        yb[o * 64 + j] = v0;
        yb[o * 64 + 32 + j] = v1;
    }
}
#endif

#if !defined(BO_WAVE_EMU)
// ---- small-batch form (BASELINE.json configs[3]: uci.py analyses ONE position, 20-block x 256 net) ---------------------
// bo_k_conv3x3 gives a board one workgroup, so a batch of 1 would use 1 of 256 CUs.  Here one board's layer is cut into
// (C_out/16) x 4 workgroups: 16 output channels x 16 positions (two board rows) each, and the four waves of a workgroup
// split K (a quarter of the input channels each, all 9 taps) and reduce through LDS: 64 workgroups for a 256-filter layer,
// 144 v_mfma_f32_16x16x4_f32 per wave = 2 us of MFMA time.  Each wave stages only what it needs (its channels, the two
// rows + halo: 4 x 10 cells) -- no workgroup-wide data, one barrier before the reduction.
// Weights: Ws[oc/16][tap][c_in/16][lane = 16*(ic&3) + (oc&15)][e] = W[oc][16*g + 4*e + (ic&3)][tap].
template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
bo_k_conv3x3_small(const float *__restrict__ x, const bo_f32x4 *__restrict__ ws, const float *__restrict__ bias,
                   const float *__restrict__ res, float *__restrict__ y, int c_in_x, int mode) {
    constexpr int CQ = CIN / 4, G = CQ / 16, SLAB = 40;  // channels per wave, weight groups per wave, cells per channel
    __shared__ float Xs[4][CQ * SLAB];
    __shared__ bo_f32x4 red[4][64];
    const int ot = blockIdx.x, pt = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    const bo_f32x4 *wl = ws + ((size_t)ot * 9 * (CIN / 16) + wave * G) * 64 + lane;  // + (tap*(CIN/16) + g)*64
    // everything this wave will read is requested up front: its 9*G weight fragments (registers) and its slab (one global
    // round trip for the layer instead of one per tap / per staging iteration)
    bo_f32x4 a[9][G];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int g = 0; g < G; g++) a[tap][g] = wl[(tap * (CIN / 16) + g) * 64];
    {   // this wave's slab: channels wave*CQ .. +CQ, board rows 2pt-1 .. 2pt+2, columns -1 .. 8
        constexpr int NST = CQ * SLAB / 64;
        const float *xb = x + ((size_t)b * c_in_x + wave * CQ) * 64;
        float st[NST];
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int i = lane + 64 * u, c = i / SLAB, cell = i - c * SLAB, r = cell / 10, col = cell - r * 10 - 1, row = 2 * pt - 1 + r;
            const bool in = row >= 0 && row < 8 && col >= 0 && col < 8 && wave * CQ + c < c_in_x;
            st[u] = in ? xb[c * 64 + row * 8 + col] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NST; u++) Xs[wave][lane + 64 * u] = st[u];
    }
    __syncthreads();
    const float *xl = &Xs[wave][kq * SLAB + (n >> 3) * 10 + (n & 7)];  // + ch4*4*SLAB + (tap/3)*10 + tap%3
    bo_f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int off = (tap / 3) * 10 + tap % 3;
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap][g][e], xl[(4 * g + e) * 4 * SLAB + off], acc, 0, 0, 0);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        const bo_f32x4 s0 = red[0][lane], s1 = red[1][lane], s2 = red[2][lane], s3 = red[3][lane];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int oc = 16 * ot + 4 * kq + r;
            const size_t o = ((size_t)b * COUT + oc) * 64 + 16 * pt + n;
            float v = ((s0[r] + s1[r]) + (s2[r] + s3[r])) + bias[oc];
            if (mode == BO_CONV_BIAS_RES_RELU) v += res[o];
            if (mode != BO_CONV_RAW_BIAS) v = v > 0.0f ? v : 0.0f;
            y[o] = v;
        }
    }
}
#endif
