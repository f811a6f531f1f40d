// betaone_amd/csrc/bo_wave.h -- wavefront (64-lane) primitives for the gfx950 kernels.
//
// One CDNA4 wavefront = 64 lanes = the 64 squares of a chess board: the tree kernels run one
// wavefront per game, the move generator runs one lane per square.  Everything cross-lane goes
// through the handful of primitives below (ballot / shuffle / LDS + barrier), which map to
// v_cmp+s_mov (ballot), ds_bpermute/DPP (shuffles) and s_barrier on gfx950.
//
// When BO_WAVE_EMU is defined (tests/wave_emulator only -- never in the product build) the same
// primitives are provided by a 64-fibre lockstep emulator so that the device code can run under
// AddressSanitizer on a CPU-only machine.  The product library is always built by hipcc for gfx950.
#pragma once
#include <stdint.h>

#if defined(BO_WAVE_EMU)
#include "wave_emu.h"
#else
#include <hip/hip_runtime.h>
#define BO_DEV __device__ __forceinline__
#define BO_DEV_NOINLINE __device__ __noinline__
// every BO_KERNEL is launched as ONE wavefront per workgroup (bo_rt.h: RT_LAUNCH): telling the compiler so gives a kernel the
// whole 512-entry vector register file instead of the 128 registers a 1024-thread workgroup could use (bo_k_step spilled 83
// vector and 212 scalar registers to scratch memory without it)
#define BO_KERNEL extern "C" __global__ __launch_bounds__(64)
#define BO_SHARED __shared__
#define BO_CONST_TABLE __device__ const

BO_DEV int bo_lane() { return (int)(threadIdx.x & 63u); }
BO_DEV int bo_block() { return (int)blockIdx.x; }
BO_DEV uint64_t bo_ballot(bool p) { return __ballot(p); }
BO_DEV void bo_sync() { __syncthreads(); }
BO_DEV int bo_shfl(int v, int src) { return __shfl(v, src, 64); }
BO_DEV int bo_shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
BO_DEV int bo_shfl_up(int v, int d) { return __shfl_up(v, d, 64); }
BO_DEV int bo_atomic_add(int *p, int v) { return atomicAdd(p, v); }
BO_DEV int bo_atomic_or(int *p, int v) { return atomicOr(p, v); }
BO_DEV uint64_t bo_bitrev64(uint64_t x) { return __builtin_bitreverse64(x); }
BO_DEV unsigned long long bo_clock() { return (unsigned long long)clock64(); }
// exchange within a 16-lane row on the VALU (DPP) instead of through the LDS crossbar (ds_bpermute):
// kind 0: lane^1, 1: lane^2, 2: mirror within 8 lanes, 3: mirror within 16 lanes -- four steps combine a row
#define BO_ROW_XCHG(v, kind) \
    __builtin_amdgcn_update_dpp(0, (v), (kind) == 0 ? 0xB1 : (kind) == 1 ? 0x4E : (kind) == 2 ? 0x141 : 0x140, 0xF, 0xF, false)
#endif
#if !defined(BO_WAVE_EMU)
// value of lane `src` where src is wave-uniform: v_readlane_b32 (an SGPR result), not a ds_bpermute round trip
BO_DEV int bo_readlane(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
#else
static inline int bo_readlane(int v, int src) { return bo_shfl(v, src); }
#endif
#if defined(BO_WAVE_EMU)
static inline int bo_row_xchg_emu(int v, int kind) {
    const int l = bo_lane();
    const int src = kind == 0 ? (l ^ 1) : kind == 1 ? (l ^ 2) : kind == 2 ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15)));
    return bo_shfl(v, src);
}
#define BO_ROW_XCHG(v, kind) bo_row_xchg_emu((v), (kind))
#endif

// A hand-over between lanes of ONE wave through LDS (one lane writes, others read later): the hardware executes a wave's LDS
// operations in order, so all that is needed is that the compiler does not move the accesses across this point.  (The
// emulator runs its lanes one after another up to the next rendezvous: there it is one.)  Call in wave-uniform control flow.
#if defined(BO_WAVE_EMU)
BO_DEV void bo_wave_sync() { bo_emu::rendezvous(); }
#else
BO_DEV void bo_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif

// ---- derived primitives (identical in both builds) ---------------------------------------------
BO_DEV float bo_shfl_f(float v, int src) { return __builtin_bit_cast(float, bo_shfl(__builtin_bit_cast(int, v), src)); }
BO_DEV float bo_shfl_xor_f(float v, int m) { return __builtin_bit_cast(float, bo_shfl_xor(__builtin_bit_cast(int, v), m)); }
BO_DEV uint64_t bo_shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)bo_shfl((int)(uint32_t)v, src), hi = (uint32_t)bo_shfl((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
BO_DEV int bo_popc64(uint64_t x) { return __builtin_popcountll(x); }
BO_DEV int bo_msb64(uint64_t x) { return 63 - __builtin_clzll(x); }  // x != 0
BO_DEV int bo_lsb64(uint64_t x) { return __builtin_ctzll(x); }       // x != 0

// wave-wide integer sum (all lanes get the total)
BO_DEV int bo_wave_sum(int v) {
    for (int m = 1; m < 64; m <<= 1) v += bo_shfl_xor(v, m);
    return v;
}
// inclusive prefix sum over lanes in DESCENDING lane order (lane 63 first): lane l gets sum_{k>=l} v_k
BO_DEV int bo_wave_scan_desc(int v) {
    int lane = bo_lane();
    for (int d = 1; d < 64; d <<= 1) {
        int o = bo_shfl(v, (lane + d) & 63);
        if (lane + d < 64) v += o;
    }
    return v;
}
// wave-wide float max / sum with a FIXED butterfly order (deterministic)
BO_DEV float bo_wave_max_f(float v) {
    for (int m = 1; m < 64; m <<= 1) { float o = bo_shfl_xor_f(v, m); v = o > v ? o : v; }
    return v;
}
BO_DEV float bo_wave_sum_f(float v) {
    for (int m = 1; m < 64; m <<= 1) v = v + bo_shfl_xor_f(v, m);
    return v;
}
