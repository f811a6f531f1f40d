// betaone_amd/csrc/bo_heads.h -- everything behind the tower's head convolutions in two launches (gfx950).
//
// PolicyValueNet.forward after the two 1x1 head convolutions (/root/reference/network.py:186-197) and the softmax the
// search applies to the logits (/root/reference/mcts.py:185,287):
//     logits = policy_fc(p)            [B,128]  x [128,4672]           p = ReLU'd policy planes, flattened
//     probs  = softmax(logits, dim=1)
//     value  = tanh(value_fc2(relu(value_fc1(v))))   [B,2048] x [2048,256], then 256 -> 1;  v = ReLU'd value planes
// As library calls this is two small GEMMs on forked streams, a softmax and a tail kernel: ~25 us of kernels that are launch-
// and latency-bound (0.6 GFLOP in total) plus ~15 us of dependency gaps between graph nodes, behind every one of the ten
// evaluations of a ply.  Here:
//   bo_k_heads_tiles   workgroups 0 .. NP-1   logits tiles on v_mfma_f32_16x16x4_f32 (K = 128), a wave = [32 boards x 32 outputs]; the four
//                                             waves of a workgroup cover [128 boards x 32 outputs] (more than 64 boards), [64 x 64] or
//                                             [32 x 128] (a cohort's 64 boards: 73 workgroups instead of 146, none of their waves idle);
//                                             every fragment of a wave is requested before its first MFMA (one memory round trip)
//                      workgroups NP ..       value_fc1 partial tile [64 boards x 64 hidden x 128 of K = 2048]: both operand
//                                             tiles are staged through LDS with coalesced loads; partial sums go to scratch
//   bo_k_heads_rows    one board per workgroup: softmax of its logits row (the row passes through registers once) and
//                      value = tanh(value_fc2(relu(sum of the 16 K chunks in a fixed order + bias)))
// (A single launch with a device-wide barrier between the two phases was measured first: what crosses the barrier has to be
// written through / fetched past the XCDs' private L2s, or every workgroup has to write back and invalidate its L2; either way
// the tiles' stores took longer than the kernel boundary costs -- profiles/r02_heads_probe.md.)
// Both products are computed transposed (M = outputs, N = boards), so that a lane's accumulator is four consecutive outputs of
// one board: 16-byte stores.  Operands need no packing: a lane's fragments of four consecutive K-steps are one float4 of a
// row of the activations / of the Linear weight (any partition of K into groups of four is a valid K-step as long as both
// operands use the same one).
#pragma once
#if !defined(BO_WAVE_EMU)
#include <hip/hip_runtime.h>
#include "bo_tower.h"

#define BO_HEADS_NA 4672
#define BO_HEADS_KP 128
#define BO_HEADS_KV 2048
#define BO_HEADS_NH 256
#define BO_HEADS_KS 16        // K chunks of value_fc1 (partial sums reduced by the rows kernel, in a fixed order)
#define BO_HEADS_PITCH 132    // floats per LDS tile row: 128 + 4 (conflict-free 16-byte fragment reads)

struct bo_heads_args {
    const void *p, *v;                        // [B,128], [B,2048]: float32, or float16 behind the fp16 tower (bo_tower_h.h)
    const float *wp, *bp, *w1, *b1, *w2, *b2;  // policy_fc [4672,128]+[4672]; value_fc1 [256,2048]+[256]; value_fc2 [256]+[1]
    float *policy_out, *value_out;            // [B,4672] (probabilities if softmax != 0, else logits), [B]
    float *vpart;                             // scratch [16 K chunks][B][256]: partial sums of value_fc1
    int B, softmax;
    int pb;                                   // boards per policy workgroup: 128, 64 or 32 (bo_heads_policy_boards)
};
static inline int bo_heads_policy_boards(int batch) { return batch > 64 ? 128 : batch > 32 ? 64 : 32; }

// four consecutive activations as float32 from a float32 or float16 array (element index e, a multiple of 4)
template <bool HALF>
__device__ __forceinline__ bo_f32x4 bo_heads_load4(const void *base, size_t e) {
    if constexpr (HALF) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 h = *reinterpret_cast<const h4 *>(reinterpret_cast<const _Float16 *>(base) + e);
        return bo_f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        return *reinterpret_cast<const bo_f32x4 *>(reinterpret_cast<const float *>(base) + e);
    }
}

template <bool HALF>
__global__ void __launch_bounds__(256)
bo_k_heads_tiles(bo_heads_args a) {
    __shared__ __attribute__((aligned(16))) float tileA[64 * BO_HEADS_PITCH], tileW[64 * BO_HEADS_PITCH];  // value tiles [64][128 + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, i = lane & 15;
    // A policy workgroup's four waves cover `pb` boards x (128 / pb) output tiles of 32: pb = 128 (one output tile, the boards of four
    // waves) down to 32 (four output tiles of the same 32 boards) -- with few boards (a cohort's 64) the launch then has half / a quarter
    // of the workgroups and no idle waves; every output element is computed by the same instruction sequence for every pb.
    const int B = a.B, pb = a.pb, nbw = pb >> 5, otw = 4 / nbw, NPT = (BO_HEADS_NA / 32 + otw - 1) / otw, n_policy = ((B + pb - 1) / pb) * NPT;
    const int wg = (int)blockIdx.x;
    if (wg < n_policy) {
        // ---- logits tile: boards pb*rb + 32*(wave % nbw) + [0,32) (N), outputs 32*(otw*ct + wave / nbw) + [0,32) (M) ----
        const int rb = wg / NPT, ct = wg - rb * NPT, r0 = pb * rb + 32 * (wave % nbw), c0 = 32 * (otw * ct + wave / nbw);
        if (c0 >= BO_HEADS_NA) return;  // (146 output tiles do not divide by four)
        const bo_f32x4 *w4 = reinterpret_cast<const bo_f32x4 *>(a.wp);
        constexpr int PG = BO_HEADS_KP / 16;
        bo_f32x4 fp[PG][2], fw[PG][2];
#pragma unroll
        for (int t = 0; t < PG; t++) {
            const int kk = 16 * t + 4 * kq;
#pragma unroll
            for (int c = 0; c < 2; c++) fw[t][c] = w4[((size_t)(c0 + 16 * c + i) * BO_HEADS_KP + kk) >> 2];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int row = r0 + 16 * rt + i;
                fp[t][rt] = row < B ? bo_heads_load4<HALF>(a.p, (size_t)row * BO_HEADS_KP + kk) : bo_f32x4{0, 0, 0, 0};
            }
        }
        bo_f32x4 bias[2];
#pragma unroll
        for (int c = 0; c < 2; c++) bias[c] = *reinterpret_cast<const bo_f32x4 *>(a.bp + c0 + 16 * c + 4 * kq);
        __builtin_amdgcn_sched_barrier(0);
        bo_f32x4 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) acc[c][rt] = bo_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < PG; t++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[t][c][e], fp[t][rt][e], acc[c][rt], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = c0 + 16 * c + 4 * kq;  // this lane: outputs col .. col+3 of board r0 + 16*rt + i
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int row = r0 + 16 * rt + i;
                if (row < B)
                    *reinterpret_cast<bo_f32x4 *>(a.policy_out + (size_t)row * BO_HEADS_NA + col) =
                        bo_f32x4{acc[c][rt][0] + bias[c][0], acc[c][rt][1] + bias[c][1], acc[c][rt][2] + bias[c][2], acc[c][rt][3] + bias[c][3]};
            }
        }
    } else {
        // ---- value_fc1 partial tile: boards 64*rbv + [0,64), hidden units 64*ht + [0,64), K chunk 128*ks + [0,128).
        //      Both operand tiles are contiguous 512-byte row segments: fetched with whole-wave coalesced loads into LDS (a
        //      fragment-order fetch of rows 8 KB apart serialises on a few L2 channels), fragments come from LDS. ----
        const int vt = wg - n_policy, ks = vt & (BO_HEADS_KS - 1), ht = (vt >> 4) & 3, rbv = vt >> 6;
        const int r0 = 64 * rbv, h0 = 64 * ht, k0 = (BO_HEADS_KV / BO_HEADS_KS) * ks;
        const bo_f32x4 *w4 = reinterpret_cast<const bo_f32x4 *>(a.w1);
        bo_f32x4 ga[8], gw[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, row = idx >> 5, c4 = idx & 31;  // 32 lanes per 512-byte row segment
            ga[q] = r0 + row < B ? bo_heads_load4<HALF>(a.v, ((size_t)(r0 + row)) * BO_HEADS_KV + k0 + 4 * c4) : bo_f32x4{0, 0, 0, 0};
            gw[q] = w4[(((size_t)(h0 + row)) * BO_HEADS_KV + k0) / 4 + c4];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, row = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<bo_f32x4 *>(&tileA[row * BO_HEADS_PITCH + 4 * c4]) = ga[q];
            *reinterpret_cast<bo_f32x4 *>(&tileW[row * BO_HEADS_PITCH + 4 * c4]) = gw[q];
        }
        __syncthreads();
        // wave: hidden units 32*(wave & 1) + [0,32) (M), boards 32*(wave >> 1) + [0,32) (N)
        const int hb = 32 * (wave & 1), bb = 32 * (wave >> 1);
        bo_f32x4 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) acc[c][rt] = bo_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < BO_HEADS_KV / BO_HEADS_KS / 16; t++) {
            bo_f32x4 fw[2], fv[2];
#pragma unroll
            for (int c = 0; c < 2; c++) fw[c] = *reinterpret_cast<const bo_f32x4 *>(&tileW[(hb + 16 * c + i) * BO_HEADS_PITCH + 16 * t + 4 * kq]);
#pragma unroll
            for (int rt = 0; rt < 2; rt++) fv[rt] = *reinterpret_cast<const bo_f32x4 *>(&tileA[(bb + 16 * rt + i) * BO_HEADS_PITCH + 16 * t + 4 * kq]);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[c][e], fv[rt][e], acc[c][rt], 0, 0, 0);
        }
        // partial sums [ks][board][256 hidden]: this lane holds hidden h0 + hb + 16c + 4kq + [0,4) of board r0 + bb + 16rt + i
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int row = r0 + bb + 16 * rt + i;
                if (row < B) *reinterpret_cast<bo_f32x4 *>(a.vpart + ((size_t)ks * B + row) * BO_HEADS_NH + h0 + hb + 16 * c + 4 * kq) = acc[c][rt];
            }
    }
}

// one board per workgroup (grid = B): the softmax of its logits row and its value
extern "C" __global__ void __launch_bounds__(256)
bo_k_heads_rows(bo_heads_args a) {
    __shared__ float sred[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = a.B, row = (int)blockIdx.x;
    // value: the K chunks' partial sums in a fixed order, bias, ReLU, value_fc2, tanh (requested first: it overlaps the row)
    float part[BO_HEADS_KS];
#pragma unroll
    for (int ks = 0; ks < BO_HEADS_KS; ks++) part[ks] = a.vpart[((size_t)ks * B + row) * BO_HEADS_NH + tid];
    bo_f32x4 *r4 = reinterpret_cast<bo_f32x4 *>(a.policy_out + (size_t)row * BO_HEADS_NA);
    constexpr int N4 = BO_HEADS_NA / 4, IT = (N4 + 255) / 256;  // 1168 float4, 5 per thread
    const float ninf = -__builtin_inff();
    bo_f32x4 x[IT];
    float mx = ninf;
    if (a.softmax) {
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int k = tid + 256 * u;
            x[u] = k < N4 ? r4[k] : bo_f32x4{ninf, ninf, ninf, ninf};
        }
    }
    float hsum = 0.0f;
#pragma unroll
    for (int ks = 0; ks < BO_HEADS_KS; ks++) hsum += part[ks];
    hsum += a.b1[tid];
    hsum = (hsum > 0.0f ? hsum : 0.0f) * a.w2[tid];
    if (a.softmax) {
#pragma unroll
        for (int u = 0; u < IT; u++) mx = fmaxf(mx, fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3])));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, m, 64)); hsum += __shfl_xor(hsum, m, 64); }
    if (lane == 0) { sred[wave] = mx; sred[4 + wave] = hsum; }
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    if (tid == 0) a.value_out[row] = tanhf(((sred[4] + sred[5]) + (sred[6] + sred[7])) + a.b2[0]);
    if (a.softmax) {
        float sum = 0.0f;
#pragma unroll
        for (int u = 0; u < IT; u++) {
            if (tid + 256 * u < N4) {
#pragma unroll
                for (int e = 0; e < 4; e++) { x[u][e] = expf(x[u][e] - mx); sum += x[u][e]; }  // (bo_tree.h: row_load_max_sum keeps this exp and this order)
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
        __syncthreads();
        if (lane == 0) sred[wave] = sum;
        __syncthreads();
        sum = (sred[0] + sred[1]) + (sred[2] + sred[3]);
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int k = tid + 256 * u;
            if (k < N4) r4[k] = bo_f32x4{x[u][0] / sum, x[u][1] / sum, x[u][2] / sum, x[u][3] / sum};
        }
    }
}

// ---- fp16 head planes, any number of rows (fast mode: 4 096 .. 131 072 rows per evaluation) -------------------------------------------
// Behind the fp16 tower (bo_tower_h.h) the head planes are fp16 and so are the Linear weights the reference's autocast evaluation
// multiplies with (network.py:186-197 under torch.autocast; softmax of mcts.py:287 in float32).  As library calls that is a GEMM that
// writes fp16 logits, a widening copy, a softmax that reads and writes the float32 rows again (9.8 GB of traffic at 131 072 rows),
// the value GEMM, a clamp, a small GEMM and a tanh.  Here the probabilities are written ONCE and nothing is read back -- a logits tile
// costs 16 MFMAs to recompute, a round trip through memory costs more:
//   bo_k_heads_policy_h<0>  workgroup = 256 boards (64 per wave, their B fragments resident in registers) x one of 10 ranges of the 146
//                           output tiles [32 outputs x 32 boards], v_mfma_f32_32x32x16_f16: running (max, sum of exp) per board over
//                           the range -> stats[range][board].  The 32 weight rows of a tile are fetched once per workgroup with
//                           coalesced loads into a double-buffered LDS tile; the four waves take their A fragments from there.
//   bo_k_heads_policy_h<1>  same decomposition: a board's ten partial (max, sum) pairs are merged, every tile is computed again and
//                           exp(x - max) / sum is stored.  Transposed product (M = outputs, N = boards): a lane's accumulator group is
//                           four consecutive outputs of one board, 16-byte stores.   <2>: the logits, no statistics.
//   bo_k_heads_value_h      64 boards per workgroup: value_fc1 over the whole K = 2048 (activation chunks staged through LDS, the wave's
//                           64 weight rows streamed from L2), bias, ReLU, value_fc2 and tanh in the epilogue -- no partial sums in memory.
// Both decompose over (board blocks x output ranges), so 4 096 rows already give 160 workgroups.
// K is taken in the order (kg, step, i) -> k = KSPAN * kg + 8 * step + i for both operands (any permutation of K is a valid product):
// a lane's fragments of consecutive steps are then consecutive 16-byte pieces of ONE row.
typedef _Float16 bo_hh8 __attribute__((ext_vector_type(8)));

#ifndef BO_HEADS_NT_STORES
#define BO_HEADS_NT_STORES 1                              // probabilities are written once and read by a later kernel: stream them past L2
#endif
#define BO_HEADS_PS 10                                    // output ranges per board block
#define BO_HEADS_PT 15                                    // tiles per range (the last range: 11)
#define BO_HEADS_WPITCH (BO_HEADS_KP + 8)                 // halves per LDS weight row: + 16 bytes (conflict-free 16-byte fragment reads)

struct bo_heads_h_args {
    const _Float16 *p, *v;        // [B,128], [B,2048]
    const _Float16 *wp, *w1;      // policy_fc.weight [4672,128], value_fc1.weight [256,2048], fp16
    const float *bp, *b1, *w2, *b2;  // float32: policy_fc.bias, value_fc1.bias, value_fc2.weight [256], value_fc2.bias
    float *policy_out, *value_out;   // [B,4672] probabilities (softmax != 0) or logits; [B]
    float *stats;                    // [BO_HEADS_PS][B][2]: (max, sum of exp(x - max)) of a board over one output range
    int B, softmax;
};

template <int PASS>
__global__ void __launch_bounds__(256)
bo_k_heads_policy_h(bo_heads_h_args a) {
    __shared__ __attribute__((aligned(16))) _Float16 wt[2][32 * BO_HEADS_WPITCH];
    __shared__ float2 merged[PASS == 1 ? 256 : 1];  // (max, 1 / sum) of the workgroup's boards
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = lane >> 5, n = lane & 31;
    const int B = a.B, r0 = 256 * (int)blockIdx.x + 64 * wave, rg = (int)blockIdx.y;
    constexpr int NT = BO_HEADS_NA / 32, KS = BO_HEADS_KP / 16;  // 146 output tiles, 8 K-steps
    const int t0 = BO_HEADS_PT * rg, t1 = t0 + BO_HEADS_PT < NT ? t0 + BO_HEADS_PT : NT;
    int row[2];
    bo_hh8 fb[2][KS];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        row[h] = r0 + 32 * h + n < B ? r0 + 32 * h + n : B - 1;  // (rows past the end: loads clamped, stores skipped)
        const bo_hh8 *prow = reinterpret_cast<const bo_hh8 *>(a.p + (size_t)row[h] * BO_HEADS_KP + 64 * kg);
#pragma unroll
        for (int s = 0; s < KS; s++) fb[h][s] = prow[s];
    }
    // staging: a tile's 32 weight rows x 256 bytes = 512 16-byte pieces, two per thread (16 threads per row: coalesced)
    const int srow = tid >> 3, sc = 2 * (tid & 7);
    const bo_hh8 *wsrc = reinterpret_cast<const bo_hh8 *>(a.wp + (size_t)srow * BO_HEADS_KP) + sc;
    constexpr size_t TSTRIDE = (size_t)32 * BO_HEADS_KP / 8;  // bo_hh8 units per tile of 32 weight rows
    bo_hh8 g0 = wsrc[t0 * TSTRIDE], g1 = wsrc[t0 * TSTRIDE + 1];
    *reinterpret_cast<bo_hh8 *>(&wt[0][srow * BO_HEADS_WPITCH + 8 * sc]) = g0;
    *reinterpret_cast<bo_hh8 *>(&wt[0][srow * BO_HEADS_WPITCH + 8 * sc + 8]) = g1;
    const bo_f32x4 *bias4 = reinterpret_cast<const bo_f32x4 *>(a.bp);
    const float ninf = -__builtin_inff();
    float m[2] = {ninf, ninf}, ssum[2] = {0.0f, 0.0f};
    if (PASS == 1) {  // merge the board's partial statistics (every range's workgroup does this for its own use)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float pm[BO_HEADS_PS], ps[BO_HEADS_PS];
#pragma unroll
            for (int q = 0; q < BO_HEADS_PS; q++) {
                const float2 st = *reinterpret_cast<const float2 *>(a.stats + 2 * ((size_t)q * B + row[h]));
                pm[q] = st.x; ps[q] = st.y;
            }
            float mm = pm[0];
#pragma unroll
            for (int q = 1; q < BO_HEADS_PS; q++) mm = fmaxf(mm, pm[q]);
            float tot = 0.0f;
#pragma unroll
            for (int q = 0; q < BO_HEADS_PS; q++) tot += ps[q] * __expf(pm[q] - mm);
            if (kg == 0) merged[64 * wave + 32 * h + n] = float2{mm, 1.0f / tot};
        }
    }
    // The write passes multiply the other way round (M = boards, N = outputs): accumulator r of a lane is output 32t + (lane & 31) of board
    // 32h + (r & 3) + 8 (r >> 2) + 4 kg, so that one store instruction writes 128 consecutive bytes of two boards' rows -- whole lines
    // (with M = outputs a lane's four stores of a tile fill one line 32 bytes at a time: 2.9 TB/s at 131 072 rows, profiles/r04_heads_f16.md).
    float bm[2][16], binv[2][16];
    if (PASS == 1) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float2 st = merged[64 * wave + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * kg];
                bm[h][r] = st.x; binv[h][r] = st.y;
            }
    }
    const bool full = r0 + 64 <= B;
    float *obase = a.policy_out + (size_t)(r0 + 4 * kg) * BO_HEADS_NA + n;
    for (int t = t0; t < t1; t++) {
        const int cur = (t - t0) & 1;
        if (t + 1 < t1) { g0 = wsrc[(t + 1) * TSTRIDE]; g1 = wsrc[(t + 1) * TSTRIDE + 1]; }
        __syncthreads();  // tile t is in wt[cur]; every wave is done with wt[cur ^ 1] (tile t - 1)
        bo_hh8 fa[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) fa[s] = *reinterpret_cast<const bo_hh8 *>(&wt[cur][n * BO_HEADS_WPITCH + 64 * kg + 8 * s]);
        bo_f32x16 acc[2];
        if (PASS == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bo_f32x4 bq = bias4[(32 * t + 8 * q + 4 * kg) >> 2];
#pragma unroll
                for (int e = 0; e < 4; e++) { acc[0][4 * q + e] = bq[e]; acc[1][4 * q + e] = bq[e]; }
            }
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int h = 0; h < 2; h++) acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], fb[h][s], acc[h], 0, 0, 0);
        } else {
            const float bn = a.bp[32 * t + n];
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[0][r] = bn; acc[1][r] = bn; }
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int h = 0; h < 2; h++) acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[h][s], fa[s], acc[h], 0, 0, 0);
        }
        if (t + 1 < t1) {
            *reinterpret_cast<bo_hh8 *>(&wt[cur ^ 1][srow * BO_HEADS_WPITCH + 8 * sc]) = g0;
            *reinterpret_cast<bo_hh8 *>(&wt[cur ^ 1][srow * BO_HEADS_WPITCH + 8 * sc + 8]) = g1;
        }
        if (PASS == 0) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float tm = acc[h][0];
#pragma unroll
                for (int r = 1; r < 16; r++) tm = fmaxf(tm, acc[h][r]);
                const float mn = fmaxf(m[h], tm);
                float add = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; r++) add += __expf(acc[h][r] - mn);
                ssum[h] = ssum[h] * __expf(m[h] - mn) + add;
                m[h] = mn;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int brow = 32 * h + (r & 3) + 8 * (r >> 2);  // (+ 4 kg: in obase)
                    const float o = PASS == 1 ? __expf(acc[h][r] - bm[h][r]) * binv[h][r] : acc[h][r];
                    if (full || r0 + 4 * kg + brow < B) {
                        float *dst = obase + (size_t)brow * BO_HEADS_NA + 32 * t;
                        if (BO_HEADS_NT_STORES) __builtin_nontemporal_store(o, dst);
                        else *dst = o;
                    }
                }
        }
    }
    if (PASS == 0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {  // the board's other sixteen outputs of every tile sit in lane ^ 32
            const float m2 = __shfl_xor(m[h], 32, 64), s2 = __shfl_xor(ssum[h], 32, 64);
            const float mm = fmaxf(m[h], m2);
            const float tot = ssum[h] * __expf(m[h] - mm) + s2 * __expf(m2 - mm);
            if (kg == 0 && r0 + 32 * h + n < B) *reinterpret_cast<float2 *>(a.stats + 2 * ((size_t)rg * B + row[h])) = float2{mm, tot};
        }
    }
}

#define BO_HEADS_VKC 256                       // halves of K per LDS chunk
#define BO_HEADS_VPITCH (BO_HEADS_VKC + 8)     // + 16 bytes: conflict-free 16-byte fragment reads

__global__ void __launch_bounds__(256)
bo_k_heads_value_h(bo_heads_h_args a) {
    __shared__ __attribute__((aligned(16))) _Float16 tileV[64 * BO_HEADS_VPITCH];
    __shared__ float red[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = lane >> 5, n = lane & 31;
    const int B = a.B, r0 = 64 * (int)blockIdx.x;
    constexpr int NC = BO_HEADS_KV / BO_HEADS_VKC, KS = BO_HEADS_VKC / 16;  // 8 chunks of 16 K-steps
    // staging: 64 rows x 512 bytes per chunk = 2048 16-byte pieces, 8 per thread (32 threads per row: coalesced)
    bo_hh8 g[8];
    auto fetch = [&](int c) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, rr = idx >> 5, c8 = idx & 31;
            const int row = r0 + rr < B ? r0 + rr : B - 1;
            g[q] = *reinterpret_cast<const bo_hh8 *>(a.v + (size_t)row * BO_HEADS_KV + BO_HEADS_VKC * c + 8 * c8);
        }
    };
    bo_f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][t][r] = 0.0f;
    // this wave: hidden units 64 * wave + 32 * mt + (lane & 31); a lane's 16 steps of a chunk are 256 consecutive bytes of its row
    const bo_hh8 *wrow[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) wrow[mt] = reinterpret_cast<const bo_hh8 *>(a.w1 + (size_t)(64 * wave + 32 * mt + n) * BO_HEADS_KV + 128 * kg);
    fetch(0);
    for (int c = 0; c < NC; c++) {
        __syncthreads();  // (the previous chunk's fragment reads are done)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, rr = idx >> 5, c8 = idx & 31;
            *reinterpret_cast<bo_hh8 *>(&tileV[rr * BO_HEADS_VPITCH + 8 * c8]) = g[q];
        }
        __syncthreads();
        if (c + 1 < NC) fetch(c + 1);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            bo_hh8 fa[2][8];
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int s = 0; s < 8; s++) fa[mt][s] = wrow[mt][(BO_HEADS_VKC * c) / 8 + 8 * half + s];
#pragma unroll
            for (int s = 0; s < 8; s++) {
                bo_hh8 fv[2];
#pragma unroll
                for (int t = 0; t < 2; t++) fv[t] = *reinterpret_cast<const bo_hh8 *>(&tileV[(32 * t + n) * BO_HEADS_VPITCH + 128 * kg + 8 * (8 * half + s)]);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int t = 0; t < 2; t++) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][s], fv[t], acc[mt][t], 0, 0, 0);
            }
        }
    }
    // value = tanh(value_fc2(relu(value_fc1 + b1)) + b2): this lane holds hidden 64*wave + 32*mt + 8*q + 4*kg + e of boards 32*t + n
    float part[2] = {0.0f, 0.0f};
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int h0 = 64 * wave + 32 * mt + 8 * q + 4 * kg;
            const bo_f32x4 b4 = *reinterpret_cast<const bo_f32x4 *>(a.b1 + h0), w4 = *reinterpret_cast<const bo_f32x4 *>(a.w2 + h0);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float h = acc[mt][t][4 * q + e] + b4[e];
                    part[t] += (h > 0.0f ? h : 0.0f) * w4[e];
                }
        }
#pragma unroll
    for (int t = 0; t < 2; t++) part[t] += __shfl_xor(part[t], 32, 64);
    if (kg == 0) { red[wave][n] = part[0]; red[wave][32 + n] = part[1]; }
    __syncthreads();
    if (tid < 64 && r0 + tid < B) a.value_out[r0 + tid] = tanhf(((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + a.b2[0]);
}
#endif
