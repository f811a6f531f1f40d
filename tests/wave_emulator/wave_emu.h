// tests/wave_emulator/wave_emu.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A 64-fibre lockstep emulator of one CDNA wavefront, enough to execute the device code of
// betaone_amd/csrc/*.h on a CPU-only machine under AddressSanitizer/UBSan (GPU sanitizers are not
// available on the MI355X pool).  Workgroups run one after another; the 64 lanes of a workgroup
// are ucontext fibres scheduled round-robin; every cross-lane primitive is a rendezvous that all
// 64 lanes must reach (the kernels only use cross-lane operations in wave-uniform control flow;
// the emulator aborts if that is ever violated).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <functional>

#define BO_DEV static inline
#define BO_DEV_NOINLINE static
#define BO_KERNEL static
#define BO_SHARED static
#define BO_CONST_TABLE static const

namespace bo_emu {
enum { WAVE = 64, STACK_BYTES = 1 << 20 };
struct Lane {
    ucontext_t ctx;
    char *stack;
    bool done, waiting;
    uint64_t slot[2];
    unsigned ncoll;
};
struct Block {
    Lane lane[WAVE];
    ucontext_t sched;
    int cur, bid, grid;
    const std::function<void()> *body;
};
inline Block *&blk() { static Block *b = nullptr; return b; }

inline void trampoline() {
    Block *b = blk();
    (*b->body)();
    b->lane[b->cur].done = true;
    swapcontext(&b->lane[b->cur].ctx, &b->sched);
}
inline void rendezvous() {
    Block *b = blk();
    Lane &l = b->lane[b->cur];
    l.waiting = true;
    swapcontext(&l.ctx, &b->sched);
}
inline void launch(int grid, const std::function<void()> &body) {
    static Block *b = nullptr;
    if (!b) {
        b = new Block();
        for (int i = 0; i < WAVE; i++) b->lane[i].stack = (char *)malloc(STACK_BYTES);
    }
    blk() = b;
    b->body = &body;
    b->grid = grid;
    for (int bid = 0; bid < grid; bid++) {
        b->bid = bid;
        for (int i = 0; i < WAVE; i++) {
            Lane &l = b->lane[i];
            getcontext(&l.ctx);
            l.ctx.uc_stack.ss_sp = l.stack;
            l.ctx.uc_stack.ss_size = STACK_BYTES;
            l.ctx.uc_link = &b->sched;
            l.done = l.waiting = false;
            l.ncoll = 0;
            makecontext(&l.ctx, (void (*)())trampoline, 0);
        }
        for (;;) {
            int n_done = 0, n_wait = 0;
            for (int i = 0; i < WAVE; i++) {
                Lane &l = b->lane[i];
                if (l.done) { n_done++; continue; }
                b->cur = i;
                l.waiting = false;
                swapcontext(&b->sched, &l.ctx);
                if (l.done) n_done++;
                else n_wait++;
            }
            if (n_done == WAVE) break;
            if (n_done != 0) {
                fprintf(stderr, "wave_emu: divergent cross-lane operation (%d lanes exited, %d waiting) in block %d\n",
                        n_done, n_wait, bid);
                abort();
            }
        }
    }
}
struct Idx { int x; };
inline int lane() { return blk()->cur; }
template <class T> inline void put(T v) {
    Lane &l = blk()->lane[blk()->cur];
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    l.slot[l.ncoll & 1] = u;
}
template <class T> inline T get(int src, unsigned coll) {
    uint64_t u = blk()->lane[src & 63].slot[coll & 1];
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
}  // namespace bo_emu

#define threadIdx (bo_emu::Idx{bo_emu::lane()})
#define blockIdx (bo_emu::Idx{bo_emu::blk()->bid})

BO_DEV int bo_lane() { return bo_emu::lane(); }
BO_DEV int bo_block() { return bo_emu::blk()->bid; }
BO_DEV void bo_sync() { bo_emu::rendezvous(); }
BO_DEV uint64_t bo_ballot(bool p) {
    bo_emu::Lane &l = bo_emu::blk()->lane[bo_emu::lane()];
    unsigned c = l.ncoll;
    bo_emu::put<uint64_t>(p ? 1 : 0);
    l.ncoll++;
    bo_emu::rendezvous();
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (bo_emu::get<uint64_t>(i, c) & 1) << i;
    return m;
}
BO_DEV int bo_shfl(int v, int src) {
    bo_emu::Lane &l = bo_emu::blk()->lane[bo_emu::lane()];
    unsigned c = l.ncoll;
    bo_emu::put<int>(v);
    l.ncoll++;
    bo_emu::rendezvous();
    return bo_emu::get<int>(src & 63, c);
}
BO_DEV int bo_shfl_xor(int v, int m) { return bo_shfl(v, bo_emu::lane() ^ m); }
BO_DEV int bo_shfl_up(int v, int d) {
    int l = bo_emu::lane();
    int o = bo_shfl(v, (l - d) & 63);
    return l - d >= 0 ? o : v;
}
BO_DEV unsigned long long bo_clock() { return 0ull; }
BO_DEV int bo_atomic_add(int *p, int v) { int o = *p; *p = o + v; return o; }
BO_DEV int bo_atomic_or(int *p, int v) { int o = *p; *p = o | v; return o; }
BO_DEV uint64_t bo_bitrev64(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
    return __builtin_bswap64(x);
}
