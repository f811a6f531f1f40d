// betaone_amd/csrc/bo_fastw.h -- FAST search mode.  NOT the reference's semantics.
//
// The reference's search has no virtual loss, so every NN batch holds one position repeated up to 96 times and the root
// never gets more than two children (mcts.py:186,203,210-254).  This mode keeps the reference's
// interfaces (game stack, legal-move order, draw rules, input planes, (state, pi, z) records) but runs a conventional
// batched AlphaZero-style search, clearly labelled as diverging from the reference:
//   * L leaves per game per step, selected one after another under a VIRTUAL LOSS; NN batch = G x L rows, row = g*L + r;
//     two descents that end in the same unexpanded leaf share its row;
//   * full-width expansion: every legal move becomes a child, prior = policy mass renormalised over the legal moves;
//     Dirichlet noise on all root priors;
//   * W of a node is the value sum seen by the player who moved INTO it, so PUCT needs no negation:
//       score = W/n + cpuct * P * sqrt(N_parent) / (1 + n);
//   * TREE REUSE: after a move the played child's subtree becomes the next search's tree (the reference rebuilds the
//     tree from scratch every move, mcts.py:176).
//
// The virtual loss is never written into the tree.  While a step's L descents run, the tree is READ-ONLY: a record's
// effective statistics are its stored ones plus the descents of this step already in flight through it,
//       n_eff = n + c,   W_eff = W - (float)c,      c = earlier descents of the step whose path holds the record
// (c comes from the step's own paths: one byte per (descent, depth) = the child index chosen there, kept in LDS; only
// descents that share the whole prefix can share the next record, so the candidates shrink level by level).  The next
// launch's backup then applies, per simulation in order and per path record, n += 1 and W += +/-v: 16 bytes per path
// node, written by the lane that owns the depth -- the parent chain of a simulation is updated in parallel, and all
// simulations of a step with one memory round trip.  Known-terminal leaves (mate / draw found at their first visit) need
// no evaluation; they are backed up with the step's other simulations.
//
// Layout for bandwidth (the select kernel that SURVEY.md section 8d prices against the HBM roofline).  A tree is a
// per-game arena of 16-byte records
//     { int32 n; float w; float prior; int32 link }
// allocated in GRANULES of BO_FW_GR (= 8) records = 128 bytes = one L2 line.  The children of a node are ONE run:
//     [ header granule(s): the node's position (80 B) + its child count ][ ceil(children / 8) record granules ]
// link = first record granule | (record granules - 1) << 24, or a negative state code.  A level of a descent is one
// coalesced request for the run's record granules only (12 B/child of statistics + the 4-B link that shares the lines
// anyway; the padding is at most 7 records -- the 32-record blocks of round 2 moved 1.75x the algorithmic bytes); the
// header is touched by leaf materialisation and re-rooting only.  Moves (2 B per record) live in a parallel array off
// the select path.  A bump allocator per game; re-rooting copies the kept subtree breadth-first into the game's second
// arena (compaction and garbage collection in one pass).
//
// One step = three launches inside the captured graph:
//   bo_k_fw_apply   one wave per (game, row): priors of the evaluated leaf -> a new run
//   bo_k_fw_select  HALF a wave per game (32 lanes = 32 records = four granules per request), several games interleaved per
//                   half-wave with every game's next run requested before any is consumed (a descent is a chain of
//                   dependent reads: bandwidth = runs in flight x bytes / latency): backup of the previous step's
//                   simulations, then L descents                                        <- the select + backup kernel
//   bo_k_fw_leaf    one wave per (game, row): make-move, legal moves, is_game_over(claim_draw=True), planes 98..119
// Arithmetic is plain binary32 in a fixed order; tests/fast_reference.py restates it in NumPy and whole trees are compared
// bit for bit (there is no reference implementation of this mode to compare with).
#pragma once
#include "bo_tree.h"

#ifndef BO_FW_GR
#define BO_FW_GR 8            // records per granule (8 x 16 B = one 128-byte L2 line)
#endif
#define BO_FW_HG ((96 + 16 * BO_FW_GR - 1) / (16 * BO_FW_GR))  // granules of a run's header
#define BO_FW_PATH_CAP 64     // deepest path of one descent
#define BO_FW_LMAX 64         // leaves per game per step
#define BO_FW_LINK_MASK 0xFFFFFF
#define BO_FW_SQRT_TAB 4096   // FastW::sqrt_tab: RN(sqrt(n)) for the visit counts a PUCT scan usually sees
#define BO_FW_RCP_TAB 256     // FastW::rcp_tab: RN(1 / k)
#define FW_UNVISITED (-1)
#define FW_MATE (-2)          // known terminal: the player who moved into the node delivered mate
#define FW_DRAW (-3)
#define FW_SIM_MATE (-2)      // sim_row codes of simulations that ended in a known terminal (no NN row)
#define FW_SIM_DRAW (-3)
enum { FW_SEL_NT = 1, FW_SEL_ROOT_IN_REGS = 2, FW_SEL_DENSE = 4, FW_SEL_LANE = 8, FW_SEL_OCT = 16, FW_SEL_QUAD = 32 };  // FastW::sel_flags

struct alignas(16) WRec {
    int n;        // visits; -1 = padding of a run's last granule, never selected
    float w;      // value sum from the point of view of the player who moved into the node
    float prior;
    int link;
};
struct alignas(16) FwHead {  // in front of a run's records
    DPos pos;     // position of the node whose children the run holds
    int nrec;     // its children
    int pad[3];
};

// Per-game control block of the fast mode: ONE contiguous row of f.CS ints per game, so that the select + backup kernel fetches a
// game's whole per-launch state with one coalesced request per half-wave (as separate [G] / [G][L] arrays it was ~25 dependent
// 4-byte reads per game: the kernel's time was its bookkeeping, not the tree).
//   [FWC_*] scalars | row_slot[L] | row_plink[L] | row_nlegal[L] | row_term[L] | row_sim[L] | sim_row[L] | sim_plen[L]
enum { FWC_NROWS = 0,   // NN rows of the step in flight
       FWC_NSTEP = 1,   // simulations of the step in flight
       FWC_CUR = 2,     // live arena (0 / 1)
       FWC_TOP = 3,     // granules in use
       FWC_LEVELS = 4, FWC_KIDS = 5, FWC_GRAN = 6, FWC_PNODES = 7, FWC_TERM = 8,  // counters: levels descended, children scanned,
                        // record granules requested (x 16 * GR = bytes the select path moved), path nodes updated by the backup (x 16 B
                        // algorithmic), terminal simulations
       FWC_HEAD = 16 };
enum { FWR_SLOT = 0,    // leaf record of row r
       FWR_PLINK = 1,   // link of the run the leaf lives in (= its parent's children)
       FWR_NLEGAL = 2, FWR_TERM = 3,
       FWR_SIM = 4,     // the simulation whose path materialises the row's position
       FWS_ROW = 5,     // row whose value simulation s backs up (FW_SIM_*: known terminal)
       FWS_PLEN = 6,    // its path length
       FWR_FIELDS = 7 };
#define FWC_F(L, field, i) (FWC_HEAD + (field) * (L) + (i))
// index of (simulation s, depth d) in a game's block of FastW::sim_path.  Depth-major: the L simulations' record ids of one depth are
// adjacent, so the depths a step's paths actually reach (~3 of 64) x L ids are ONE 128-byte line per game for the backup to fetch and
// for the descents to write -- simulation-major they were L lines 256 bytes apart.
#define FW_PIDX(L, s, d) ((size_t)(d) * (size_t)(L) + (size_t)(s))

struct FastW {
    int L, NG, CS;             // leaves per game per step; granules per arena; ints per control block (a multiple of 32)
    int sel_ut, sel_flags;     // bo_k_fw_select: games per half-wave (1, 2 or 4); FW_SEL_*
    WRec *arena;               // [G][2][NG * GR]   (two arenas per game, side by side: re-rooting compacts from one into the other)
    bo_mv *amove;              // [G][2][NG * GR]   move of each record
    int *ctl;                  // [G][CS] control blocks
    const float *sqrt_tab, *rcp_tab;  // [BO_FW_SQRT_TAB] RN(sqrt(n)); [BO_FW_RCP_TAB] RN(1 / k), entry 0 = 0
    DPos *row_pos;             // [G][L] position of row r's leaf
    bo_mv *row_moves;          // [G][L][256] its legal moves (python-chess order)
    int *sim_path;             // [G][PATH_CAP][L] record ids root..leaf, DEPTH-major (record id = granule * GR + index): FW_PIDX
    int *played_now;           // [G] move played by the last bo_k_play (0: none); the same array as Eng::played_now
};

BO_DEV int *fw_ctl(const FastW &f, int g) { return f.ctl + (size_t)g * f.CS; }
BO_DEV size_t fw_arena_off(const FastW &f, int g, int which) { return ((size_t)g * 2 + (size_t)which) * (size_t)f.NG * BO_FW_GR; }
BO_DEV WRec *fw_arena(const FastW &f, int g) { return f.arena + fw_arena_off(f, g, fw_ctl(f, g)[FWC_CUR]); }
BO_DEV bo_mv *fw_moves(const FastW &f, int g) { return f.amove + fw_arena_off(f, g, fw_ctl(f, g)[FWC_CUR]); }
BO_DEV int fw_first(int link) { return link & BO_FW_LINK_MASK; }
BO_DEV int fw_ngran(int link) { return ((link >> 24) & 127) + 1; }
BO_DEV int fw_link(int first, int ngran) { return first | ((ngran - 1) << 24); }
BO_DEV int fw_gran_for(int n) { return (n + BO_FW_GR - 1) / BO_FW_GR; }
BO_DEV const FwHead *fw_head(const WRec *A, int link) { return reinterpret_cast<const FwHead *>(A + (size_t)(fw_first(link) - BO_FW_HG) * BO_FW_GR); }
BO_DEV const FwHead *fw_head_at(const WRec *A, int gran) { return reinterpret_cast<const FwHead *>(A + (size_t)gran * BO_FW_GR); }

// one record as a single 16-byte request (optionally non-temporal: a run is read once per launch)
#if defined(BO_WAVE_EMU)
template <bool NT> BO_DEV WRec fw_ld(const WRec *p) { return *p; }
struct fw_nw { int n; float w; };
BO_DEV fw_nw fw_ld_nw(const WRec *p) { fw_nw r; r.n = p->n; r.w = p->w; return r; }
BO_DEV void fw_st_nw(WRec *p, int n, float w) { p->n = n; p->w = w; }
#else
typedef int fw_i4 __attribute__((ext_vector_type(4)));
typedef int fw_i2 __attribute__((ext_vector_type(2)));
template <bool NT> BO_DEV WRec fw_ld(const WRec *p) {
    const fw_i4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const fw_i4 *>(p)) : *reinterpret_cast<const fw_i4 *>(p);
    // (elements into scalars first: __builtin_bit_cast applied to a vector ELEMENT read element 0 whatever the index -- clang 19 /
    //  ROCm 7.2 -- so every score ignored the priors; found by the lock-step probe against the emulator build)
    const int i0 = v[0], i1 = v[1], i2 = v[2], i3 = v[3];
    WRec r;
    r.n = i0; r.w = __builtin_bit_cast(float, i1); r.prior = __builtin_bit_cast(float, i2); r.link = i3;
    return r;
}
struct fw_nw { int n; float w; };
BO_DEV fw_nw fw_ld_nw(const WRec *p) {
    const fw_i2 v = *reinterpret_cast<const fw_i2 *>(p);
    const int i0 = v[0], i1 = v[1];
    fw_nw r;
    r.n = i0; r.w = __builtin_bit_cast(float, i1);
    return r;
}
BO_DEV void fw_st_nw(WRec *p, int n, float w) {
    fw_i2 v;
    v[0] = n; v[1] = __builtin_bit_cast(int, w);
    *reinterpret_cast<fw_i2 *>(p) = v;
}
#endif

// lane-strided partial sums + butterfly: the summation order tests/fast_reference.py mirrors
BO_DEV float fw_sum(const float *v, int n) {
    float a = 0.0f;
    for (int j = bo_lane(); j < n; j += 64) a = a + v[j];
    return bo_wave_sum_f(a);
}

struct FwShared {
    bo_mv moves[BO_MAX_MOVES];
    bo_mv moves2[BO_MAX_MOVES];
    float pv[BO_MAX_MOVES];
    ChainBuf chain;
};

// granules the run of a row's leaf takes (header + records), 0 for a terminal leaf; `top` advances only when it fits --
// bo_k_fw_apply (one wave per row) and bo_k_fw_select (the allocator's owner) apply the same rule in the same order
BO_DEV int fw_row_need(int term, int nlegal) { return term == 0 ? BO_FW_HG + fw_gran_for(nlegal) : 0; }

// ---- apply: the evaluated leaf of row r gets its children ------------------------------------------------------------
BO_KERNEL void bo_k_fw_apply(Eng e, FastW f, const float *policy, int kind) {
    BO_SHARED float pv[BO_MAX_MOVES];
    const int L = f.L, g = bo_block() / L, r = bo_block() % L, lane = bo_lane();
    const int *ctl = fw_ctl(f, g);
    if (e.phase[g] != PH_RUN || r >= ctl[FWC_NROWS]) return;
    const size_t ro = (size_t)g * L + r;
    WRec *A = fw_arena(f, g);
    const int slot = ctl[FWC_F(L, FWR_SLOT, r)], t = ctl[FWC_F(L, FWR_TERM, r)];
    if (t > 0) {  // found terminal at its first visit: remember it, no children
        if (lane == 0) A[slot].link = t == 1 ? FW_MATE : FW_DRAW;
        return;
    }
    const int n = ctl[FWC_F(L, FWR_NLEGAL, r)];
    const bo_mv *mv = f.row_moves + ro * BO_MAX_MOVES;
    const float *prow = policy + ro * BO_NUM_ACTIONS;
    if (kind == POLICY_PROBS) {
        for (int j = lane; j < n; j += 64) pv[j] = prow[move_to_index(mv[j])];
    } else {  // softmax over the legal moves only
        float mx = -__builtin_inff();
        for (int j = lane; j < n; j += 64) { const float x = prow[move_to_index(mv[j])]; pv[j] = x; mx = x > mx ? x : mx; }
        mx = bo_wave_max_f(mx);
        for (int j = lane; j < n; j += 64) pv[j] = bo_expf(pv[j] - mx);
    }
    bo_sync();
    const float sum = fw_sum(pv, n);
    for (int j = lane; j < n; j += 64) pv[j] = sum > 0.0f ? pv[j] / sum : 1.0f / (float)n;
    bo_sync();
    if (slot == 0 && e.c.use_noise) {  // Dirichlet noise on every root prior
        const double *nz = e.noise + (size_t)g * BO_MAX_MOVES;
        for (int j = lane; j < n; j += 64) {
            const float a = e.c.keep * pv[j];
            pv[j] = (float)((double)a + e.c.eps * nz[j]);
        }
        bo_sync();
    }
    // this row's run starts behind the runs of the rows before it (rows of one game are applied by different waves)
    int first = ctl[FWC_TOP];
    for (int q = 0; q < r; q++) {
        const int need = fw_row_need(ctl[FWC_F(L, FWR_TERM, q)], ctl[FWC_F(L, FWR_NLEGAL, q)]);
        if (first + need <= f.NG) first += need;  // (a run that does not fit is refused; bo_k_fw_select advances `top` by the same rule)
    }
    const int ngran = fw_gran_for(n);
    if (first + BO_FW_HG + ngran > f.NG) {  // arena full: the leaf stays unexpanded (its value is still backed up)
        if (lane == 0) bo_atomic_or(&e.status[g], ST_NODE_OVERFLOW);
        return;
    }
    bo_mv *M = fw_moves(f, g);
    const size_t r0 = (size_t)(first + BO_FW_HG) * BO_FW_GR;
    for (int i = lane; i < ngran * BO_FW_GR; i += 64) {
        WRec c;
        c.n = i < n ? 0 : -1; c.w = 0.0f; c.prior = i < n ? pv[i] : 0.0f; c.link = FW_UNVISITED;
        A[r0 + i] = c;
        M[r0 + i] = i < n ? mv[i] : (bo_mv)0;
    }
    if (lane == 0) {
        FwHead *h = reinterpret_cast<FwHead *>(A + (size_t)first * BO_FW_GR);
        h->pos = f.row_pos[ro];
        h->nrec = n; h->pad[0] = h->pad[1] = h->pad[2] = 0;
        A[slot].link = fw_link(first + BO_FW_HG, ngran);
    }
}

// ---- select + backup ----------------------------------------------------------------------------------------------------
// first maximum in child order within a half-wave: (score, index) through four DPP row rounds and one cross-row exchange
#define BO_FW_ARGMAX(os_expr, oi_expr)                                                      \
    {                                                                                       \
        const float os = (os_expr);                                                         \
        const int oi = (oi_expr);                                                           \
        if (os > best || (os == best && oi < bi)) { best = os; bi = oi; }                   \
    }

template <int LCAP> struct FwMask { typedef unsigned long long T; };
template <> struct FwMask<4> { typedef unsigned T; };
template <> struct FwMask<8> { typedef unsigned T; };
template <> struct FwMask<16> { typedef unsigned T; };
BO_DEV int fw_ctz(unsigned m) { return __builtin_ctz(m); }
BO_DEV int fw_ctz(unsigned long long m) { return __builtin_ctzll(m); }

// Half a wave per game, UT games interleaved per half-wave.  NT: runs below the root are requested non-temporally (read once per
// launch); the root's run, read by every descent of the step, takes the cached path.  ROOTC: the first 64 records of the root's
// run stay in registers for the whole launch.
//   1. every game's control block, the step's values and the Eng scalars: one round trip for all 2 * UT games of the wave, into LDS
//   2. backup of the step in flight: lane c owns path depth c (a record sits at one depth only, so all updates of a record are
//      made by ONE lane, in simulation order: no cross-lane ordering is needed); the paths of BO_FW_BK_CH simulations of every
//      game are fetched together, then their records, and simulations that share a record are chained in registers -- a step's
//      backup costs three dependent round trips for the whole wave, not one per simulation and game
//   3. L descents per game over the read-only tree; rows / simulations are noted in the LDS copy of the control block
//   4. the control blocks go back with one coalesced store per game
// W lanes per game (32: half a wave, two records per lane; 8: an eighth, RPL = 5 records per lane and 8 games per instruction),
// UT games interleaved per group of W lanes
template <int W, int RPL, int UT, int LCAP, bool NT, bool ROOTC>
BO_DEV void fw_select_body(const Eng &e, const FastW &f, const float *value, int kind) {
    typedef typename FwMask<LCAP>::T mask_t;
    constexpr int NGRP = 64 / W, NSL = NGRP * UT;  // groups of lanes per wave; games per wave
    constexpr int CSL = FWC_HEAD + FWR_FIELDS * LCAP, NCR = (CSL + W - 1) / W, NVR = (LCAP + W - 1) / W;
    static_assert(LCAP <= W * 2, "row list scan covers two words per lane");
    static_assert(RPL <= 16, "in-flight counts are packed for sixteen passes");
    constexpr int BO_FW_BK_CH = UT >= 4 ? 2 : 4;  // simulations of a game whose backups are fetched together and chained in registers
    BO_SHARED int s_ctl[NSL][NCR * W];                         // the games' control blocks
    BO_SHARED float s_val[NSL][NVR * W];                       // values of the step's rows
    constexpr int NW = (LCAP + 3) / 4;
    BO_SHARED unsigned s_idx[NSL][BO_FW_PATH_CAP][NW];         // child index chosen at depth d by descent s of the step: byte s of the depth's words
    BO_SHARED float s_rcp[BO_FW_RCP_TAB];                          // RN(1 / k)
    const int lane = bo_lane(), half = lane / W, c = lane % W, hb = half * W;  // (`half`: this lane's group)
    const unsigned long long gmask = W == 64 ? ~0ull : ((1ull << W) - 1ull);
    const int L = f.L, S = e.c.S, G = e.c.G;
    const float cpuct = e.c.cpuct;
    // the 2 * UT games of a workgroup are consecutive: their arenas are addressed as one uniform base + a 32-bit offset
    const int g0 = bo_block() * NSL;
    WRec *const base = f.arena + fw_arena_off(f, g0, 0);
    unsigned aoff[UT];  // records from `base` to the game's live arena
    int sims[UT], n_rows[UT], n_step[UT], root_n[UT], root_link[UT];
    int link[UT], pn[UT], d[UT];
    mask_t M[UT];
    bool on[UT], busy[UT], done[UT];
    WRec rr[ROOTC ? UT : 1][RPL];
#define FW_G(u) (g0 + half * UT + (u))
#define FW_A(u) (base + aoff[u])
#define FW_SLOT(u) (half * UT + (u))
#define FW_C(u, i) s_ctl[FW_SLOT(u)][(i)]

    // (bo_debug_profile: shader cycles of this wave's phases, added up per workgroup row: [0] control blocks + backup, [1] the wait for
    //  the backup's stores, [2] descents, [3] tail, [4] level-loop iterations, [5] waves)
    const unsigned long long t_0 = e.c.profile ? bo_clock() : 0ull;
    unsigned long long t_1 = 0ull, t_2 = 0ull, t_3 = 0ull;
    int n_iter = 0;
    // ---- 1. control blocks, values, Eng scalars -> LDS: one round trip for the whole wave --------------------------------
    for (int i = lane; i < BO_FW_RCP_TAB; i += 64) s_rcp[i] = f.rcp_tab[i];
    {
        int creg[UT][NCR], ph[UT], top0[UT];
        float vreg[UT][NVR];
        // (requested with the control blocks, before anyone knows whether they are needed: both arenas' root records and the first
        //  BO_FW_BK_CH paths' record ids at this lane's depth -- the backup then starts one round trip earlier)
        WRec root2[UT][2];
        int pid[UT][BO_FW_BK_CH];
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            const bool in = FW_G(u) < G;
            const int gg = in ? FW_G(u) : g0;
            const int *ctl = fw_ctl(f, gg);
            BO_UNROLL
            for (int a = 0; a < 2; a++) root2[u][a] = fw_ld<false>(f.arena + fw_arena_off(f, gg, a));
            BO_UNROLL
            for (int j = 0; j < BO_FW_BK_CH; j++) pid[u][j] = f.sim_path[(size_t)gg * L * BO_FW_PATH_CAP + FW_PIDX(L, j < L ? j : L - 1, c)];
            BO_UNROLL
            for (int k = 0; k < NCR; k++) creg[u][k] = c + W * k < f.CS ? ctl[c + W * k] : 0;
            BO_UNROLL
            for (int k = 0; k < NVR; k++) vreg[u][k] = (kind != POLICY_NONE && c + W * k < L) ? value[(size_t)gg * L + c + W * k] : 0.0f;
            ph[u] = in ? e.phase[gg] : PH_IDLE;
            sims[u] = e.sims_done[gg];
            done[u] = e.root_term[gg] != 0;
        }
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            BO_UNROLL
            for (int k = 0; k < NCR; k++) FW_C(u, c + W * k) = creg[u][k];
            BO_UNROLL
            for (int k = 0; k < NVR; k++) s_val[FW_SLOT(u)][c + W * k] = vreg[u][k];
        }
        bo_wave_sync();
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            const int gg = FW_G(u) < G ? FW_G(u) : g0;
            n_rows[u] = FW_C(u, FWC_NROWS); n_step[u] = FW_C(u, FWC_NSTEP); top0[u] = FW_C(u, FWC_TOP);
            on[u] = ph[u] == PH_RUN && !(n_rows[u] > 0 && kind == POLICY_NONE);  // (rows waiting for an evaluation that has not been made)
            aoff[u] = (unsigned)(((size_t)(gg - g0) * 2 + (size_t)FW_C(u, FWC_CUR)) * (size_t)f.NG * BO_FW_GR);
            busy[u] = false;
            link[u] = -1; pn[u] = 1; d[u] = 1; M[u] = 0;
            root_n[u] = 0; root_link[u] = -1;
        }
        // ---- 2. the previous step's rows have been applied: back their values up ------------------------------------------
        WRec rootrec[UT];
        bool bk[UT];
        int maxstep = 0;
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            rootrec[u] = FW_C(u, FWC_CUR) ? root2[u][1] : root2[u][0];  // (its n is not touched by the backup below: the root only counts visits)
            if (!on[u]) { rootrec[u].n = 0; rootrec[u].link = FW_UNVISITED; }
            bk[u] = on[u] && n_rows[u] > 0;
            if (bk[u] && n_step[u] > maxstep) maxstep = n_step[u];
        }
        int term_sims[UT], deep[UT];
        BO_UNROLL
        for (int u = 0; u < UT; u++) term_sims[u] = deep[u] = 0;
        constexpr int DS = W >= 8 ? 1 : 2;  // path depths a lane owns in the register-chained backup: c (+ W with four lanes per game)
        for (int s0 = 0; s0 < maxstep; s0 += BO_FW_BK_CH) {
            int rec[UT][BO_FW_BK_CH][DS];
            bool val[UT][BO_FW_BK_CH][DS];
            float sgn[UT][BO_FW_BK_CH][DS];
            fw_nw x[UT][BO_FW_BK_CH][DS];
            BO_UNROLL
            for (int u = 0; u < UT; u++) {
                const int *sp = f.sim_path + (size_t)FW_G(u) * L * BO_FW_PATH_CAP;
                BO_UNROLL
                for (int j = 0; j < BO_FW_BK_CH; j++) {
                    const int s = s0 + j;
                    const bool ok = bk[u] && s < n_step[u];
                    const int q = ok ? FW_C(u, FWC_F(L, FWS_ROW, s)) : FW_SIM_DRAW;
                    const int plen = ok ? FW_C(u, FWC_F(L, FWS_PLEN, s)) : 0;
                    float v;
                    if (q >= 0) {  // value[] is from the leaf's side to move; the player who moved into the leaf sees -v; a found terminal has its exact value
                        const int t = FW_C(u, FWC_F(L, FWR_TERM, q));
                        v = t > 0 ? (t == 1 ? 1.0f : 0.0f) : -s_val[FW_SLOT(u)][q];
                        term_sims[u] += (ok && t > 0) ? 1 : 0;
                    } else {
                        v = q == FW_SIM_MATE ? 1.0f : 0.0f;
                        term_sims[u] += ok ? 1 : 0;
                    }
                    if (plen > W * DS) deep[u] = 1;
                    BO_UNROLL
                    for (int ds = 0; ds < DS; ds++) {
                        const int cd = c + W * ds;
                        val[u][j][ds] = cd >= 1 && cd < plen;  // the root (depth 0) only counts visits
                        sgn[u][j][ds] = ((plen - 1 - cd) & 1) ? -v : v;
                        rec[u][j][ds] = val[u][j][ds] ? ((s0 == 0 && ds == 0) ? pid[u][j] : sp[FW_PIDX(L, s, cd)]) : 0;
                    }
                }
            }
            BO_UNROLL
            for (int u = 0; u < UT; u++) {
                BO_UNROLL
                for (int j = 0; j < BO_FW_BK_CH; j++)
                    BO_UNROLL
                    for (int ds = 0; ds < DS; ds++) {
                        x[u][j][ds].n = 0; x[u][j][ds].w = 0.0f;
                        if (val[u][j][ds]) x[u][j][ds] = fw_ld_nw(FW_A(u) + rec[u][j][ds]);
                    }
            }
            BO_UNROLL
            for (int u = 0; u < UT; u++) {
                BO_UNROLL
                for (int ds = 0; ds < DS; ds++) {  // (a record sits at one depth only: the slots are independent chains)
                    BO_UNROLL
                    for (int j = 0; j < BO_FW_BK_CH; j++) {
                        if (!val[u][j][ds]) continue;
                        BO_UNROLL
                        for (int i = 0; i < j; i++)
                            if (val[u][i][ds] && rec[u][i][ds] == rec[u][j][ds]) x[u][j][ds] = x[u][i][ds];  // (the latest earlier simulation through the same record wins)
                        x[u][j][ds].n = x[u][j][ds].n + 1;
                        x[u][j][ds].w = x[u][j][ds].w + sgn[u][j][ds];
                    }
                    BO_UNROLL
                    for (int j = 0; j < BO_FW_BK_CH; j++)
                        if (val[u][j][ds]) fw_st_nw(FW_A(u) + rec[u][j][ds], x[u][j][ds].n, x[u][j][ds].w);
                }
            }
        }
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            root_n[u] = rootrec[u].n; root_link[u] = rootrec[u].link;
            if (!bk[u]) continue;
            if (deep[u]) {  // path depths W * DS .. 63 (rare): one simulation at a time
                const int *sp = f.sim_path + (size_t)FW_G(u) * L * BO_FW_PATH_CAP;
                for (int s = 0; s < n_step[u]; s++) {
                    const int q = FW_C(u, FWC_F(L, FWS_ROW, s)), plen = FW_C(u, FWC_F(L, FWS_PLEN, s));
                    float v;
                    if (q >= 0) { const int t = FW_C(u, FWC_F(L, FWR_TERM, q)); v = t > 0 ? (t == 1 ? 1.0f : 0.0f) : -s_val[FW_SLOT(u)][q]; }
                    else v = q == FW_SIM_MATE ? 1.0f : 0.0f;
                    for (int k = c + W * DS; k < plen; k += W) {
                        WRec *R = FW_A(u) + sp[FW_PIDX(L, s, k)];
                        const fw_nw y = fw_ld_nw(R);
                        fw_st_nw(R, y.n + 1, y.w + (((plen - 1 - k) & 1) ? -v : v));
                    }
                }
            }
            int top = top0[u];
            for (int q = 0; q < n_rows[u]; q++) {
                const int need = fw_row_need(FW_C(u, FWC_F(L, FWR_TERM, q)), FW_C(u, FWC_F(L, FWR_NLEGAL, q)));
                if (top + need <= f.NG) top += need;  // (bo_k_fw_apply refused the runs that do not fit, in the same order)
            }
            if (FW_C(u, FWC_F(L, FWR_SLOT, 0)) == 0 && n_step[u] == 0) root_n[u] = 1;  // the root's own evaluation counts as its first visit
            root_n[u] += n_step[u];
            sims[u] += n_step[u];
            n_rows[u] = n_step[u] = 0;
            if (c == 0) { FW_C(u, FWC_TOP) = top; FW_C(u, FWC_TERM) += term_sims[u]; }
        }
    }
    if (e.c.profile) t_1 = bo_clock();
    bo_sync();  // the backup's stores are complete before any descent reads the tree
    if (e.c.profile) t_2 = bo_clock();

    // ---- 3. up to L descents per game; the tree is read-only from here on ------------------------------------------------
    // A launch lasts as long as ONE wave's chain of dependent levels (all waves run side by side), so the chain is kept short:
    //   * what the next request needs lives in registers (link, visits of the node being expanded, depth, in-flight mask, the
    //     root's link and visits, descents made / allowed); what only the bookkeeping needs lives in LDS;
    //   * software pipeline: the moment a game's winner is known, the run behind it (or the root's, for the game's next descent) is
    //     requested -- BEFORE the level's bookkeeping (path, in-flight bytes, row list, path store), which then runs under the
    //     memory latency, as do the other games' levels (vmcnt counts in order: a game waits only for its own run);
    //   * branch-free: predicated arithmetic and clamped addresses (as divergent if-blocks a level compiled to ~850 issued
    //     instructions, two thirds of them exec-mask bookkeeping, and every conditional load was waited for inside its own block).
    enum { ST_LEVELS = 0, ST_GRAN, ST_KIDS, ST_WORDS };
    BO_SHARED int s_st[NSL][ST_WORDS];
#define FW_ST(u, i) s_st[FW_SLOT(u)][(i)]
    int nmax[UT];  // descents this launch may make: min(L, S - sims)
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        if (c == 0) {
            FW_ST(u, ST_LEVELS) = 0; FW_ST(u, ST_GRAN) = 0; FW_ST(u, ST_KIDS) = 0;
            FW_C(u, FWC_NROWS) = n_rows[u]; FW_C(u, FWC_NSTEP) = n_step[u];
        }
        nmax[u] = S - sims[u] < L ? S - sims[u] : L;
        if (!on[u]) continue;
        if (done[u] || (root_link[u] >= 0 && sims[u] >= S)) {
            done[u] = true;
        } else if (root_link[u] < 0) {  // root not expanded yet: its evaluation is row 0 (no simulation attached)
            if (c == 0) {
                FW_C(u, FWC_F(L, FWR_SLOT, 0)) = 0; FW_C(u, FWC_F(L, FWR_PLINK, 0)) = -1; FW_C(u, FWC_F(L, FWR_SIM, 0)) = -1;
                FW_C(u, FWC_NROWS) = 1;
            }
        } else {
            busy[u] = true;
            link[u] = root_link[u]; pn[u] = root_n[u] + 1;
            if (c == 0) f.sim_path[(size_t)FW_G(u) * L * BO_FW_PATH_CAP] = 0;
            if (ROOTC) {
                const int nrec = fw_ngran(root_link[u]) * BO_FW_GR;
                const WRec *R = FW_A(u) + (size_t)fw_first(root_link[u]) * BO_FW_GR;
                WRec pad; pad.n = -1; pad.w = 0.0f; pad.prior = 0.0f; pad.link = FW_UNVISITED;
                BO_UNROLL
                for (int k = 0; k < RPL; k++) rr[u][k] = c + W * k < nrec ? fw_ld<false>(R + c + W * k) : pad;
                if (c == 0) FW_ST(u, ST_GRAN) = fw_ngran(root_link[u]);
            }
        }
    }
    bo_wave_sync();
    WRec r[UT][RPL];
    float sqt[UT];
    // request the run behind link[u] (unconditional loads from clamped addresses: a lane beyond the run re-reads its first record,
    // the same cache line; an idle game reads granule 0 of its arena, always there) and the square root of the visits
#define BO_FW_ISSUE(u)                                                                                                  \
    {                                                                                                                   \
        const bool cached_ = ROOTC && d[u] == 1;                                                                        \
        const int lk_ = (busy[u] && !cached_) ? link[u] : 0;                                                            \
        const int nrec_ = fw_ngran(lk_) * BO_FW_GR;                                                                     \
        const WRec *R_ = FW_A(u) + (size_t)fw_first(lk_) * BO_FW_GR;                                                    \
        BO_UNROLL                                                                                                       \
        for (int k_ = 0; k_ < RPL; k_++) r[u][k_] = fw_ld<NT>(R_ + (c + W * k_ < nrec_ ? c + W * k_ : 0));              \
        sqt[u] = f.sqrt_tab[pn[u] < BO_FW_SQRT_TAB ? pn[u] : 0];                                                        \
    }
    bool any = false;
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        any = any || busy[u];
        BO_FW_ISSUE(u)
    }
    while (bo_ballot(any) != 0) {
        any = false;
        n_iter++;
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            const int slot = FW_SLOT(u);
            const bool lv = busy[u];
            const int lk = lv ? link[u] : 0;
            const int ngran = fw_ngran(lk), nrec = ngran * BO_FW_GR, first = fw_first(lk);
            const int dd = d[u] < BO_FW_PATH_CAP ? d[u] : BO_FW_PATH_CAP - 1;
            const bool atroot = d[u] == 1;
            if (ROOTC && atroot) {
                BO_UNROLL
                for (int k = 0; k < RPL; k++) r[u][k] = rr[u][k];
            }
            // descents of this step in flight through this lane's candidates: those that share the whole path so far (mask M) and
            // chose this lane's child at this depth (one byte per descent and depth, LCAP bytes = NW words per depth)
            unsigned iw[NW];
            BO_UNROLL
            for (int w = 0; w < NW; w++) iw[w] = s_idx[slot][dd][w];
            // a candidate's PUCT score from its statistics with the descents in flight through it: q + u,
            //   q = W_eff * rcp(n_eff)   u = (cpuct * P * sqrt(N)) * rcp(1 + n_eff)     rcp(k) = RN(1 / k)
            // rcp and sqrt come from tables; counts beyond them (rare) take ONE wave-uniform side path with the exact operations
            int cnt[RPL], ne[RPL];
            bool ok[RPL];
            float rq[RPL], ru[RPL];
            bool beyond = lv && pn[u] >= BO_FW_SQRT_TAB;
            unsigned cpk = 0u, cpk1 = 0u;  // descents in flight through this lane's records: 4 bits per pass (an index b belongs to lane b % W,
                                           // pass b / W); passes 8..15 (a lane of the four-lane form holds ten records) in the second word
            BO_UNROLL
            for (int sp = 0; sp < LCAP - 1; sp++) {
                const unsigned bb = (iw[sp >> 2] >> (8 * (sp & 3))) & 255u;
                const unsigned mine = (unsigned)((M[u] >> sp) & 1) & ((bb % W) == (unsigned)c ? 1u : 0u);
                cpk += (bb / W < 8u ? mine : 0u) << (4 * (bb / W & 7u));
                if (RPL > 8) cpk1 += ((bb / W >= 8u && bb / W < 16u) ? mine : 0u) << (4 * (bb / W & 7u));
            }
            int nkl = 0;
            BO_UNROLL
            for (int k = 0; k < RPL; k++) {
                if (LCAP <= 16) {
                    cnt[k] = k < 8 ? (int)((cpk >> (4 * k)) & 15u) : k < 16 ? (int)((cpk1 >> (4 * (k - 8))) & 15u) : 0;
                } else {  // (more than 15 descents can share a record: count them one by one)
                    cnt[k] = 0;
                    BO_UNROLL
                    for (int sp = 0; sp < LCAP - 1; sp++)
                        cnt[k] += (int)((M[u] >> sp) & 1) & ((int)((iw[sp >> 2] >> (8 * (sp & 3))) & 255u) == c + W * k ? 1 : 0);
                }
                ok[k] = lv && c + W * k < nrec && r[u][k].n >= 0;
                nkl += ok[k] ? 1 : 0;
                ne[k] = r[u][k].n + cnt[k];
                const int t = ne[k] < 0 ? 0 : ne[k] > BO_FW_RCP_TAB - 2 ? BO_FW_RCP_TAB - 2 : ne[k];
                rq[k] = s_rcp[t]; ru[k] = s_rcp[t + 1];
                beyond = beyond || (ok[k] && ne[k] > BO_FW_RCP_TAB - 2);
            }
            float sq = sqt[u];
            if (bo_ballot(beyond) != 0) {
                if (pn[u] >= BO_FW_SQRT_TAB) sq = sqrtf((float)pn[u]);
                BO_UNROLL
                for (int k = 0; k < RPL; k++)
                    if (ne[k] > BO_FW_RCP_TAB - 2) { rq[k] = 1.0f / (float)ne[k]; ru[k] = 1.0f / (float)(1 + ne[k]); }
            }
#define BO_FW_SCORE(rec, cnt, ne, rq, ru, ok, sc)                                                               \
            float sc;                                                                                           \
            {                                                                                                   \
                const float we = (rec).w - (float)(cnt);                                                        \
                const float t1_ = cpuct * (rec).prior;                                                          \
                const float t2_ = t1_ * sq;                                                                     \
                const float uu = t2_ * (ru);                                                                    \
                const float qv = (ne) > 0 ? we * (rq) : 0.0f;                                                   \
                sc = qv + uu;                                                                                   \
                sc = ((ok) && sc == sc) ? sc : -__builtin_inff();                                               \
            }
            float best = -__builtin_inff();
            int bi = c, bne = ne[0], bl = r[u][0].link;
            BO_UNROLL
            for (int k = 0; k < RPL; k++) {  // (a later record of the lane wins only with a strictly better score: first maximum in child order)
                BO_FW_SCORE(r[u][k], cnt[k], ne[k], rq[k], ru[k], ok[k], sck)
                const bool better = k == 0 || sck > best;
                best = better ? sck : best; bi = better ? c + W * k : bi; bne = better ? ne[k] : bne; bl = better ? r[u][k].link : bl;
            }
            const bool wide = bo_ballot(lv && nrec > W * RPL) != 0;  // a run with more records than one pass takes, somewhere in the wave
            int extra = 0;
            if (wide) {
                if (lv && nrec > W * RPL) {
                    const WRec *R = FW_A(u) + (size_t)first * BO_FW_GR;
                    for (int i = W * RPL + c; i < nrec; i += W) {
                        const WRec rx = fw_ld<false>(R + i);
                        int cntx = 0;
                        BO_UNROLL
                        for (int sp = 0; sp < LCAP - 1; sp++)
                            cntx += (int)((M[u] >> sp) & 1) & ((int)((iw[sp >> 2] >> (8 * (sp & 3))) & 255u) == i ? 1 : 0);
                        const bool okx = rx.n >= 0;
                        const int nex = rx.n + cntx;
                        const float rqx = nex > 0 ? 1.0f / (float)nex : 0.0f, rux = 1.0f / (float)(1 + nex);
                        extra += okx ? 1 : 0;
                        BO_FW_SCORE(rx, cntx, nex, rqx, rux, okx, scx)
                        if (scx > best) { best = scx; bi = i; bne = nex; bl = rx.link; }
                    }
                }
            }
#undef BO_FW_SCORE
            // first maximum in child order within the half-wave: the maximum by four DPP row rounds and one cross-row exchange, then
            // the lowest child index among the lanes that hold it (a lane's index is lane + 32 * pass: lowest pass first, then lowest lane)
            float mx = best;
            int nk = nkl;  // children scanned at this level: the lanes' counts added up alongside
#define BO_FW_RED(kind_)                                                                                                 \
            { const float o = __builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, mx), kind_)); mx = o > mx ? o : mx; nk += BO_ROW_XCHG(nk, kind_); }
            BO_FW_RED(0) BO_FW_RED(1)
            if (W >= 8) BO_FW_RED(2)
            if (W >= 16) BO_FW_RED(3)
            if (W >= 32) { const float o = bo_shfl_xor_f(mx, 16); mx = o > mx ? o : mx; nk += bo_shfl_xor(nk, 16); }
#undef BO_FW_RED
            // the lowest child index among the lanes that hold the maximum: a minimum over the group
            int win = (lv && best == mx && mx > -__builtin_inff()) ? bi : 0x7fffffff;
#define BO_FW_MIN(kind_) { const int o = BO_ROW_XCHG(win, kind_); win = o < win ? o : win; }
            BO_FW_MIN(0) BO_FW_MIN(1)
            if (W >= 8) BO_FW_MIN(2)
            if (W >= 16) BO_FW_MIN(3)
            if (W >= 32) { const int o = bo_shfl_xor(win, 16); win = o < win ? o : win; }
#undef BO_FW_MIN
            if (win == 0x7fffffff) win = -1;
            const bool nan_all = lv && win < 0;  // every score was NaN: take the first child (it exists: a run is never empty)
            if (bo_ballot(nan_all) != 0) {
                if (nan_all) {
                    win = 0;
                    if (c == 0) { bi = 0; bne = r[u][0].n; bl = r[u][0].link; bo_atomic_or(&e.status[FW_G(u)], ST_NAN_SCORE); }  // (no in-flight visits added: the search is broken anyway)
                }
            }
            // the winner's statistics come from the lane that scored it (index win belongs to lane win & 31, whose own best it is)
            const int src = hb + (win & (W - 1));
            const int w_ne = bo_shfl(bne, src);
            int w_l = bo_shfl(bl, src);
            // ---- what this game requests next, and the request itself --------------------------------------------------------
            const int old_link = link[u], old_d = d[u], s = n_step[u];
            const bool over = lv && w_l >= 0 && old_d + 1 >= BO_FW_PATH_CAP;  // path buffer full: the visit counts as a draw
            if (over) w_l = FW_DRAW;
            const bool cont = lv && w_l >= 0, ended = lv && !cont, more = ended && s + 1 < nmax[u];
            mask_t Mn = M[u];
            BO_UNROLL
            for (int sp = 0; sp < LCAP - 1; sp++)  // earlier descents that went elsewhere no longer share the path
                Mn &= ~((mask_t)((int)((iw[sp >> 2] >> (8 * (sp & 3))) & 255u) != win ? 1 : 0) << sp);
            link[u] = cont ? w_l : more ? root_link[u] : link[u];
            pn[u] = cont ? w_ne + 1 : root_n[u] + s + 2;  // (a new descent: the root's visits, the step's earlier descents and this one)
            d[u] = cont ? old_d + 1 : 1;
            M[u] = cont ? Mn : (mask_t)(((mask_t)1 << (s + 1)) - 1);
            n_step[u] = ended ? s + 1 : s;
            busy[u] = cont || more;
            BO_FW_ISSUE(u)
            any = any || busy[u];
            // ---- the level's bookkeeping, under the latency of the request above -------------------------------------------
            const int leaf = first * BO_FW_GR + win;
            if (lv && c == 0) {
                reinterpret_cast<unsigned char *>(&s_idx[slot][dd][0])[s] = (unsigned char)win;
                f.sim_path[(size_t)FW_G(u) * L * BO_FW_PATH_CAP + FW_PIDX(L, s, dd)] = leaf;  // (the path, level by level: read back by the next launch's backup)
                FW_ST(u, ST_LEVELS) += 1; FW_ST(u, ST_KIDS) += nk;
                if (!(ROOTC && atroot)) FW_ST(u, ST_GRAN) += ngran;
            }
            if (wide && extra) bo_atomic_add(&FW_ST(u, ST_KIDS), extra);
            if (bo_ballot(over) != 0) { if (over && c == 0) bo_atomic_or(&e.status[FW_G(u)], ST_DEPTH_OVERFLOW); }
            bo_wave_sync();  // lane 0's words above are read by the other lanes below and in later iterations
            if (bo_ballot(ended) != 0) {  // ---- end of a descent
                // a leaf another descent of this step already selected shares that descent's row
                // (every lane reads the words lane 0 rewrites below BEFORE the ballots: lock-step on the GPU, and the emulator's lanes
                //  run one after another between two rendezvous)
                const bool fresh = ended && w_l == FW_UNVISITED;
                const int n_rows_l = FW_C(u, FWC_NROWS);
                const uint64_t h0 = bo_ballot(fresh && c < n_rows_l && FW_C(u, FWC_F(L, FWR_SLOT, c < LCAP ? c : 0)) == leaf);
                const uint64_t h1 = LCAP > W ? bo_ballot(fresh && c + W < n_rows_l && FW_C(u, FWC_F(L, FWR_SLOT, c + W < LCAP ? c + W : 0)) == leaf) : 0ull;
                if (ended) {
                    const unsigned long long m0 = (h0 >> hb) & gmask, m1 = (h1 >> hb) & gmask;
                    const int plen = old_d + 1;
                    int q;
                    if (w_l == FW_MATE) q = FW_SIM_MATE;
                    else if (w_l == FW_DRAW) q = FW_SIM_DRAW;
                    else if (m0) q = __builtin_ctzll(m0);
                    else if (m1) q = W + __builtin_ctzll(m1);
                    else {  // becomes NN row n_rows
                        q = n_rows_l;
                        if (c == 0) {
                            FW_C(u, FWC_F(L, FWR_SLOT, q)) = leaf; FW_C(u, FWC_F(L, FWR_PLINK, q)) = old_link; FW_C(u, FWC_F(L, FWR_SIM, q)) = s;
                            FW_C(u, FWC_NROWS) = q + 1;
                        }
                    }
                    if (c == 0) { FW_C(u, FWC_F(L, FWS_ROW, s)) = q; FW_C(u, FWC_F(L, FWS_PLEN, s)) = plen; FW_C(u, FWC_NSTEP) = s + 1; }
                    if (more && c == 0) f.sim_path[(size_t)FW_G(u) * L * BO_FW_PATH_CAP + FW_PIDX(L, s + 1, 0)] = 0;  // (the next descent's path starts at the root)
                }
                bo_wave_sync();  // (the row / simulation lists in LDS)
            }
        }
    }
#undef BO_FW_ISSUE
    if (e.c.profile) t_3 = bo_clock();

    // ---- a step of known-terminal hits only needs no evaluation: account for it now ---------------------------------------
    bool allterm = false;
    BO_UNROLL
    for (int u = 0; u < UT; u++) allterm = allterm || (on[u] && FW_C(u, FWC_NROWS) == 0 && n_step[u] > 0);
    if (bo_ballot(allterm) != 0) {
        bo_sync();  // the paths written above are complete
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            const int ns = n_step[u];
            const bool at = on[u] && FW_C(u, FWC_NROWS) == 0 && ns > 0;
            if (at) {
                const int *sp = f.sim_path + (size_t)FW_G(u) * L * BO_FW_PATH_CAP;
                for (int s = 0; s < ns; s++) {  // (every simulation of the step hit a known terminal: exact values, no rows)
                    const int q = FW_C(u, FWC_F(L, FWS_ROW, s)), plen = FW_C(u, FWC_F(L, FWS_PLEN, s));
                    const float v = q == FW_SIM_MATE ? 1.0f : 0.0f;
                    for (int k = c; k < plen; k += W) {
                        if (k == 0) continue;
                        WRec *R = FW_A(u) + sp[FW_PIDX(L, s, k)];
                        const fw_nw y = fw_ld_nw(R);
                        fw_st_nw(R, y.n + 1, y.w + (((plen - 1 - k) & 1) ? -v : v));
                    }
                }
                root_n[u] += ns; sims[u] += ns; n_step[u] = 0;
                if (sims[u] >= S) done[u] = true;
            }
            bo_wave_sync();  // (every lane has read the words lane 0 rewrites now)
            if (at && c == 0) { FW_C(u, FWC_TERM) += ns; FW_C(u, FWC_PNODES) += ns; FW_C(u, FWC_NSTEP) = 0; }  // (their path nodes: one root each + the levels counted below)
        }
    }
    // ---- 4. the control blocks go back ------------------------------------------------------------------------------------
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        if (on[u] && c == 0) {
            const int gg = FW_G(u), nr = FW_C(u, FWC_NROWS);
            FW_A(u)[0].n = root_n[u];
            e.sims_done[gg] = sims[u]; e.phase[gg] = done[u] ? PH_DONE : PH_RUN;
            e.req_node[gg] = nr > 0 ? FW_C(u, FWC_F(L, FWR_SLOT, 0)) : -1;
            FW_C(u, FWC_LEVELS) += FW_ST(u, ST_LEVELS); FW_C(u, FWC_KIDS) += FW_ST(u, ST_KIDS);
            FW_C(u, FWC_GRAN) += FW_ST(u, ST_GRAN);
            FW_C(u, FWC_PNODES) += FW_ST(u, ST_LEVELS) + n_step[u];  // a path holds the root and one node per level
        }
    }
    bo_wave_sync();
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        if (!on[u]) continue;
        int *ctl = fw_ctl(f, FW_G(u));
        BO_UNROLL
        for (int k = 0; k < NCR; k++)
            if (c + W * k < FWC_HEAD + FWR_FIELDS * L) ctl[c + W * k] = FW_C(u, c + W * k);
    }
    if (e.c.profile && lane == 0) {
        unsigned long long *pp = e.prof + (size_t)g0 * BO_PROF_SLOTS;
        const unsigned long long t_4 = bo_clock();
        pp[0] += t_1 - t_0; pp[1] += t_2 - t_1; pp[2] += t_3 - t_2; pp[3] += t_4 - t_3; pp[4] += (unsigned long long)n_iter; pp[5] += 1ull;
    }
#undef FW_ST
#undef FW_G
#undef FW_A
#undef FW_SLOT
#undef FW_C
}
#undef BO_FW_ARGMAX

// ---- select + backup, ONE LANE PER GAME (leaves_per_step == 4) ---------------------------------------------------------------
// Measured on the half-wave form above (bo_debug_profile, 32768 games x 4 descents on trees of 800 simulations per move): its
// time is instruction issue, not memory -- ~250 issued instructions per game and level (cross-lane reductions, ballots, LDS
// hand-overs between lanes, exec-mask bookkeeping) of which each serves two games.  Here a lane owns a game: no cross-lane
// operation at all, a level's records are scored in a straight line of ~30 instructions each that serve 64 games at once, 64
// independent chains of dependent reads per wave, and -- every access to a game's tree coming from one lane -- program order alone
// keeps the backup's stores in front of the descents' loads.  A lane reads a run as 16-byte records of its own 128-byte
// granules: 64 lanes = 64 different lines per instruction, every line requested once (a granule's eight loads are issued back
// to back and merge in the L1's miss queue).  The control block (L = 4: every row / simulation list is one 16-byte vector) is
// read with eleven loads at once and lives in registers; the backup fetches all paths, then all records, then updates.
// The level loop is flat: every pass scores the next BO_FW_LANE_CH granules of each lane's current run, branch-free (table
// reciprocals; a count beyond the table -- rare -- makes the lane redo the pass with exact operations); a lane whose run is
// finished picks the winner, notes the level (path, in-flight byte) and goes on to the next run / the next descent at once,
// whatever the other lanes are doing.  Same arithmetic and results as the half-wave form (same tests).
#define BO_FW_LANE_CH 4  // granules (x 8 records x 4 registers) scored per pass
#if defined(BO_WAVE_EMU)
struct fw_v4 { int v[4]; int operator[](int i) const { return v[i]; } int &operator[](int i) { return v[i]; } };
BO_DEV fw_v4 fw_ld4(const int *p) { fw_v4 r; for (int i = 0; i < 4; i++) r.v[i] = p[i]; return r; }
BO_DEV void fw_st4(int *p, const fw_v4 &x) { for (int i = 0; i < 4; i++) p[i] = x.v[i]; }
#else
typedef fw_i4 fw_v4;
BO_DEV fw_v4 fw_ld4(const int *p) { return *reinterpret_cast<const fw_i4 *>(p); }
BO_DEV void fw_st4(int *p, const fw_v4 &x) { *reinterpret_cast<fw_i4 *>(p) = x; }
#endif
BO_DEV int fw_pick(const fw_v4 &x, int i) { const int a = x[0], b = x[1], c = x[2], d = x[3]; return i == 0 ? a : i == 1 ? b : i == 2 ? c : d; }
BO_DEV void fw_put(fw_v4 &x, int i, int val) {
    const int a = x[0], b = x[1], c = x[2], d = x[3];
    x[0] = i == 0 ? val : a; x[1] = i == 1 ? val : b; x[2] = i == 2 ? val : c; x[3] = i == 3 ? val : d;
}

template <bool NT>
BO_DEV void fw_select_lane_body(const Eng &e, const FastW &f, const float *value, int kind) {
    constexpr int LC = 4, NREC = BO_FW_LANE_CH * BO_FW_GR;
    BO_SHARED unsigned s_idx[BO_FW_PATH_CAP][64];  // [depth][lane]: byte s = the child index descent s of the step chose at that depth
    BO_SHARED float s_rcp[BO_FW_RCP_TAB];          // RN(1 / k)
    const int lane = bo_lane(), S = e.c.S;
    const int g = bo_block() * 64 + lane;
    const unsigned long long t_0 = e.c.profile ? bo_clock() : 0ull;
    for (int i = lane; i < BO_FW_RCP_TAB; i += 64) s_rcp[i] = f.rcp_tab[i];
    bo_sync();
    const int gg = g < e.c.G ? g : 0;
    int *ctl = fw_ctl(f, gg);
    // the control block: head | counters | .. | row_slot | row_plink | row_nlegal | row_term | row_sim | sim_row | sim_plen
    const fw_v4 c_head = fw_ld4(ctl + 0), c_stat = fw_ld4(ctl + 4), c_t = fw_ld4(ctl + 8);
    fw_v4 c_slot = fw_ld4(ctl + FWC_F(LC, FWR_SLOT, 0)), c_plink = fw_ld4(ctl + FWC_F(LC, FWR_PLINK, 0));
    const fw_v4 c_nlegal = fw_ld4(ctl + FWC_F(LC, FWR_NLEGAL, 0)), c_term = fw_ld4(ctl + FWC_F(LC, FWR_TERM, 0));
    fw_v4 c_rsim = fw_ld4(ctl + FWC_F(LC, FWR_SIM, 0)), c_srow = fw_ld4(ctl + FWC_F(LC, FWS_ROW, 0)), c_splen = fw_ld4(ctl + FWC_F(LC, FWS_PLEN, 0));
    const int ph = e.phase[gg];
    int sims = e.sims_done[gg];
    bool done = e.root_term[gg] != 0;
    float val4[LC] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (kind != POLICY_NONE) {
        BO_UNROLL
        for (int q = 0; q < LC; q++) val4[q] = value[(size_t)gg * LC + q];
    }
    int n_rows = c_head[FWC_NROWS], n_step = c_head[FWC_NSTEP], top = c_head[FWC_TOP], term_total = c_t[0];
    const bool on = g < e.c.G && ph == PH_RUN && !(n_rows > 0 && kind == POLICY_NONE);  // (rows waiting for an evaluation that has not been made)
    WRec *A = f.arena + fw_arena_off(f, gg, c_head[FWC_CUR]);
    int *sp = f.sim_path + (size_t)gg * LC * BO_FW_PATH_CAP;
    int levels = 0, grans = 0, kids = 0, n_iter = 0, root_n = 0, root_link = FW_UNVISITED;
    unsigned long long t_1 = 0ull;
    if (on) {
        const WRec root = fw_ld<false>(A);
        root_n = root.n; root_link = root.link;
        // ---- the previous step's rows have been applied: back their values up ------------------------------------------------
        if (n_rows > 0) {
            int term_sims = 0, maxlen = 0;
            float v[LC];
            int plen[LC];
            BO_UNROLL
            for (int q = 0; q < LC; q++) {
                const int need = fw_row_need(c_term[q], c_nlegal[q]);
                if (q < n_rows && top + need <= f.NG) top += need;  // (bo_k_fw_apply refused the runs that do not fit, in the same order)
            }
            BO_UNROLL
            for (int s = 0; s < LC; s++) {
                const bool ok = s < n_step;
                const int q = ok ? c_srow[s] : FW_SIM_DRAW;
                plen[s] = ok ? c_splen[s] : 0;
                if (q >= 0) {  // value[] is from the leaf's side to move; the player who moved into the leaf sees -v; a found terminal has its exact value
                    const int t = fw_pick(c_term, q);
                    const float vq = q == 0 ? val4[0] : q == 1 ? val4[1] : q == 2 ? val4[2] : val4[3];
                    v[s] = t > 0 ? (t == 1 ? 1.0f : 0.0f) : -vq;
                    term_sims += (ok && t > 0) ? 1 : 0;
                } else {
                    v[s] = q == FW_SIM_MATE ? 1.0f : 0.0f;
                    term_sims += ok ? 1 : 0;
                }
                maxlen = plen[s] > maxlen ? plen[s] : maxlen;
            }
            // eight depths at a time: the record ids of all simulations (one 16-byte read per depth: the path rows are depth-major, LC = 4
            // simulations side by side), then their (n, w), then the updates -- three round trips; simulations that share a record at a
            // depth are chained in registers
            for (int k0 = 0; k0 < maxlen; k0 += 8) {
                fw_v4 pd[8];
                BO_UNROLL
                for (int kk = 0; kk < 8; kk++) pd[kk] = fw_ld4(sp + FW_PIDX(LC, 0, k0 + kk));
                int rec[8][LC];
                bool vl[8][LC];
                fw_nw x[8][LC];
                BO_UNROLL
                for (int kk = 0; kk < 8; kk++) {
                    BO_UNROLL
                    for (int s = 0; s < LC; s++) {
                        vl[kk][s] = k0 + kk >= 1 && k0 + kk < plen[s];  // the root (depth 0) only counts visits
                        rec[kk][s] = vl[kk][s] ? pd[kk][s] : 0;
                    }
                }
                BO_UNROLL
                for (int kk = 0; kk < 8; kk++) {
                    BO_UNROLL
                    for (int s = 0; s < LC; s++) x[kk][s] = fw_ld_nw(A + rec[kk][s]);  // (an invalid slot re-reads record 0: the root's line)
                }
                BO_UNROLL
                for (int kk = 0; kk < 8; kk++) {
                    BO_UNROLL
                    for (int s = 0; s < LC; s++) {
                        BO_UNROLL
                        for (int i = 0; i < s; i++)
                            if (vl[kk][i] && vl[kk][s] && rec[kk][i] == rec[kk][s]) x[kk][s] = x[kk][i];  // (the latest earlier simulation through the same record wins)
                        x[kk][s].n = x[kk][s].n + 1;
                        x[kk][s].w = x[kk][s].w + (((plen[s] - 1 - (k0 + kk)) & 1) ? -v[s] : v[s]);
                    }
                    BO_UNROLL
                    for (int s = 0; s < LC; s++)
                        if (vl[kk][s]) fw_st_nw(A + rec[kk][s], x[kk][s].n, x[kk][s].w);
                }
            }
            if (c_slot[0] == 0 && n_step == 0) root_n = 1;  // the root's own evaluation counts as its first visit
            root_n += n_step;
            sims += n_step;
            n_rows = n_step = 0;
            term_total += term_sims;
        }
    }
    if (e.c.profile) t_1 = bo_clock();
    if (on) {
        // ---- up to L descents; this lane's loads below see its stores above (one thread, program order) ----------------------
        const int nmax = S - sims < LC ? S - sims : LC;
        bool busy = false;
        if (done || (root_link >= 0 && sims >= S)) {
            done = true;
        } else if (root_link < 0) {  // root not expanded yet: its evaluation is row 0 (no simulation attached)
            c_slot[0] = 0; c_plink[0] = -1; c_rsim[0] = -1;
            n_rows = 1;
        } else {
            busy = true;
        }
        // state of the descent in progress
        int link = root_link, pn = root_n + 1, d = 1, pass0 = 0;  // pass0: first record of the current pass within the run
        unsigned M = 0u;
        float best = -__builtin_inff(), sq = 0.0f;
        int bi = -1, bne = 0, bl = FW_UNVISITED, fl0 = -1, fl1 = -1, fl2 = -1;  // in-flight child indices at this node (-1: none)
        bool fresh_run = true;
        if (busy) sp[0] = 0;
        const float cpuct = e.c.cpuct;
        while (busy) {
            n_iter++;
            const int first = fw_first(link), ngran = fw_ngran(link), nrec = ngran * BO_FW_GR;
            if (fresh_run) {  // a new node: the descents in flight through its children, the square root of its visits
                const unsigned iw = s_idx[d][lane];
                fl0 = (M & 1u) ? (int)(iw & 255u) : -1;
                fl1 = (M & 2u) ? (int)((iw >> 8) & 255u) : -1;
                fl2 = (M & 4u) ? (int)((iw >> 16) & 255u) : -1;
                sq = pn < BO_FW_SQRT_TAB ? f.sqrt_tab[pn] : sqrtf((float)pn);
                best = -__builtin_inff(); bi = -1; bne = 0; bl = FW_UNVISITED;
                fresh_run = false;
            }
            // this pass's granules, all requested before the first is used (a record beyond the run: the run's last one again)
            WRec r[NREC];
            const WRec *R = A + (size_t)first * BO_FW_GR;
            const int nrem = nrec - pass0;  // records of the run from this pass on
            BO_UNROLL
            for (int j = 0; j < NREC; j++) r[j] = fw_ld<NT>(R + pass0 + (j < nrem ? j : nrem - 1));
            const int fa = fl0 - pass0, fb = fl1 - pass0, fc = fl2 - pass0;
            const float best_in = best;
            const int bi_in = bi, bne_in = bne, bl_in = bl;
            bool rare = false;
            BO_UNROLL
            for (int j = 0; j < NREC; j++) {
                const bool ok = j < nrem && r[j].n >= 0;
                const int cnt = (fa == j ? 1 : 0) + (fb == j ? 1 : 0) + (fc == j ? 1 : 0);
                const int ne = r[j].n + cnt;
                const int t = ne < 0 ? 0 : ne > BO_FW_RCP_TAB - 2 ? BO_FW_RCP_TAB - 2 : ne;
                const float rq = s_rcp[t], ru = s_rcp[t + 1];
                rare = rare || (ok && ne > BO_FW_RCP_TAB - 2);
                // q + u,  q = W_eff * rcp(n_eff),  u = (cpuct * P * sqrt(N)) * rcp(1 + n_eff),  rcp(k) = RN(1 / k)
                const float we = r[j].w - (float)cnt;
                const float t1 = cpuct * r[j].prior;
                const float t2 = t1 * sq;
                const float uu = t2 * ru;
                const float qv = ne > 0 ? we * rq : 0.0f;
                const float sc = qv + uu;
                const bool better = ok && sc > best;  // first maximum in child order (NaN never wins)
                kids += ok ? 1 : 0;
                best = better ? sc : best; bi = better ? pass0 + j : bi; bne = better ? ne : bne; bl = better ? r[j].link : bl;
            }
            if (rare) {  // a visit count beyond the reciprocal table: the pass again with exact operations
                best = best_in; bi = bi_in; bne = bne_in; bl = bl_in;
                for (int j = 0; j < NREC && j < nrem; j++) {
                    const WRec rx = fw_ld<false>(R + pass0 + j);
                    if (rx.n < 0) continue;
                    const int cnt = (fa == j ? 1 : 0) + (fb == j ? 1 : 0) + (fc == j ? 1 : 0);
                    const int ne = rx.n + cnt;
                    const float rq = ne > 0 ? 1.0f / (float)ne : 0.0f, ru = 1.0f / (float)(1 + ne);
                    const float we = rx.w - (float)cnt;
                    const float t1 = cpuct * rx.prior;
                    const float t2 = t1 * sq;
                    const float uu = t2 * ru;
                    const float qv = ne > 0 ? we * rq : 0.0f;
                    const float sc = qv + uu;
                    if (sc > best) { best = sc; bi = pass0 + j; bne = ne; bl = rx.link; }
                }
            }
            pass0 += NREC;
            if (pass0 < nrec) continue;  // more of this run
            // ---- the level is decided -------------------------------------------------------------------------------------
            if (bi < 0) {  // every score was NaN: take the first child (it exists: a run is never empty)
                const WRec c0 = fw_ld<false>(R);
                bi = 0; bne = c0.n; bl = c0.link;  // (no in-flight visits added: the search is broken anyway)
                bo_atomic_or(&e.status[g], ST_NAN_SCORE);
            }
            const int s = n_step, leaf = first * BO_FW_GR + bi;
            {   // this descent's byte at this depth; earlier descents that went elsewhere no longer share the path
                const unsigned iw = s_idx[d][lane];
                if ((int)(iw & 255u) != bi) M &= ~1u;
                if ((int)((iw >> 8) & 255u) != bi) M &= ~2u;
                if ((int)((iw >> 16) & 255u) != bi) M &= ~4u;
                s_idx[d][lane] = (iw & ~(255u << (8 * s))) | ((unsigned)bi << (8 * s));
            }
            sp[FW_PIDX(LC, s, d)] = leaf;
            levels++; grans += ngran;
            const int old_link = link;
            int nl = bl;
            if (nl >= 0 && d + 1 >= BO_FW_PATH_CAP) {  // path buffer full: the visit counts as a draw
                nl = FW_DRAW;
                bo_atomic_or(&e.status[g], ST_DEPTH_OVERFLOW);
            }
            pass0 = 0; fresh_run = true;
            if (nl >= 0) { link = nl; pn = bne + 1; d++; continue; }  // one level down
            // ---- end of a descent ----------------------------------------------------------------------------------------
            int q;
            if (nl == FW_MATE) q = FW_SIM_MATE;
            else if (nl == FW_DRAW) q = FW_SIM_DRAW;
            else {
                q = -1;
                BO_UNROLL
                for (int k = LC - 1; k >= 0; k--)  // a leaf another descent of this step already selected shares that descent's row
                    if (k < n_rows && c_slot[k] == leaf) q = k;
                if (q < 0) {  // becomes NN row n_rows
                    q = n_rows;
                    fw_put(c_slot, q, leaf); fw_put(c_plink, q, old_link); fw_put(c_rsim, q, s);
                    n_rows++;
                }
            }
            fw_put(c_srow, s, q); fw_put(c_splen, s, d + 1);
            n_step = s + 1;
            if (n_step < nmax) {  // the game's next descent starts at the root
                link = root_link; pn = root_n + n_step + 1; d = 1;
                M = (1u << n_step) - 1u;
                sp[FW_PIDX(LC, n_step, 0)] = 0;
            } else {
                busy = false;
            }
        }
        // ---- a step of known-terminal hits only needs no evaluation: account for it now (program order: the paths are this lane's own stores)
        const int pn_add = levels + n_step;  // a path holds the root and one node per level
        if (n_rows == 0 && n_step > 0) {
            for (int s = 0; s < n_step; s++) {
                const int q = fw_pick(c_srow, s), plen = fw_pick(c_splen, s);
                const float v = q == FW_SIM_MATE ? 1.0f : 0.0f;
                for (int k = 1; k < plen; k++) {
                    WRec *R = A + sp[FW_PIDX(LC, s, k)];
                    const fw_nw y = fw_ld_nw(R);
                    fw_st_nw(R, y.n + 1, y.w + (((plen - 1 - k) & 1) ? -v : v));
                }
            }
            term_total += n_step;
            root_n += n_step; sims += n_step; n_step = 0;
            if (sims >= S) done = true;
        }
        A[0].n = root_n;
        e.sims_done[g] = sims; e.phase[g] = done ? PH_DONE : PH_RUN;
        e.req_node[g] = n_rows > 0 ? c_slot[0] : -1;
        // the control block goes back (row_nlegal / row_term are the leaf kernel's to write)
        fw_v4 h = c_head, st = c_stat, tt = c_t;
        h[FWC_NROWS] = n_rows; h[FWC_NSTEP] = n_step; h[FWC_TOP] = top;
        st[0] = c_stat[0] + levels; st[1] = c_stat[1] + kids; st[2] = c_stat[2] + grans; st[3] = c_stat[3] + (levels ? pn_add : 0);
        tt[0] = term_total;
        fw_st4(ctl + 0, h); fw_st4(ctl + 4, st); fw_st4(ctl + 8, tt);
        fw_st4(ctl + FWC_F(LC, FWR_SLOT, 0), c_slot); fw_st4(ctl + FWC_F(LC, FWR_PLINK, 0), c_plink); fw_st4(ctl + FWC_F(LC, FWR_SIM, 0), c_rsim);
        fw_st4(ctl + FWC_F(LC, FWS_ROW, 0), c_srow); fw_st4(ctl + FWC_F(LC, FWS_PLEN, 0), c_splen);
    }
    if (e.c.profile && on) {  // per lane here: [0] control block + backup, [2] descents, [4] passes of the level loop, [5] lanes
        unsigned long long *pp = e.prof + (size_t)g * BO_PROF_SLOTS;
        const unsigned long long t_2 = bo_clock();
        pp[0] += t_1 - t_0; pp[2] += t_2 - t_1; pp[4] += (unsigned long long)n_iter; pp[5] += 1ull;
    }
}
#if defined(BO_WAVE_EMU)
#define BO_FW_LANE_OCC
#else
#define BO_FW_LANE_OCC __attribute__((amdgpu_waves_per_eu(1, 2)))
#endif
BO_FW_LANE_OCC BO_KERNEL void bo_k_fw_select_lane(Eng e, FastW f, const float *value, int kind) { fw_select_lane_body<false>(e, f, value, kind); }
BO_FW_LANE_OCC BO_KERNEL void bo_k_fw_select_lane_nt(Eng e, FastW f, const float *value, int kind) { fw_select_lane_body<true>(e, f, value, kind); }

// One instantiation per (games per half-wave, leaves-per-step capacity of the LDS path bytes, FW_SEL_* flags).  FW_SEL_DENSE:
// the register allocation is capped so that one more wave fits per SIMD (more runs in flight per CU, some spills).
#if defined(BO_WAVE_EMU)
#define BO_FW_OCC(w)
#else
#define BO_FW_OCC(w) __attribute__((amdgpu_waves_per_eu(w)))
#endif
#define BO_FW_SELECT_KERNEL(UT, LCAP, FL, OCC)                                                                         \
    OCC BO_KERNEL void bo_k_fw_select_u##UT##l##LCAP##_##FL(Eng e, FastW f, const float *value, int kind) {            \
        fw_select_body<32, 2, UT, LCAP, ((FL) & FW_SEL_NT) != 0, ((FL) & FW_SEL_ROOT_IN_REGS) != 0>(e, f, value, kind); \
    }
#define BO_FW_SELECT_FLAGSETS(UT, LCAP, DENSE_W)                                                                       \
    BO_FW_SELECT_KERNEL(UT, LCAP, 0, ) BO_FW_SELECT_KERNEL(UT, LCAP, 1, ) BO_FW_SELECT_KERNEL(UT, LCAP, 2, ) BO_FW_SELECT_KERNEL(UT, LCAP, 3, ) \
    BO_FW_SELECT_KERNEL(UT, LCAP, 4, BO_FW_OCC(DENSE_W)) BO_FW_SELECT_KERNEL(UT, LCAP, 5, BO_FW_OCC(DENSE_W))           \
    BO_FW_SELECT_KERNEL(UT, LCAP, 6, BO_FW_OCC(DENSE_W)) BO_FW_SELECT_KERNEL(UT, LCAP, 7, BO_FW_OCC(DENSE_W))
BO_FW_SELECT_FLAGSETS(4, 4, 4)   // 128 registers: 4 waves per SIMD, 128 games per CU
BO_FW_SELECT_FLAGSETS(2, 4, 6)   // 80 registers: 6 waves per SIMD, 96 games per CU
BO_FW_SELECT_KERNEL(4, 8, 0, ) BO_FW_SELECT_KERNEL(4, 8, 3, )
BO_FW_SELECT_KERNEL(2, 16, 0, ) BO_FW_SELECT_KERNEL(2, 16, 3, )
BO_FW_SELECT_KERNEL(1, 64, 0, ) BO_FW_SELECT_KERNEL(1, 64, 3, )
// eight lanes per game: eight games per wave-instruction, five records per lane
#define BO_FW_SELECT_OCT(LCAP, FL)                                                                                     \
    BO_KERNEL void bo_k_fw_select_o8l##LCAP##_##FL(Eng e, FastW f, const float *value, int kind) {                      \
        fw_select_body<8, 5, 1, LCAP, ((FL) & FW_SEL_NT) != 0, ((FL) & FW_SEL_ROOT_IN_REGS) != 0>(e, f, value, kind);   \
    }
BO_FW_SELECT_OCT(4, 0) BO_FW_SELECT_OCT(4, 1) BO_FW_SELECT_OCT(4, 2) BO_FW_SELECT_OCT(4, 3)
BO_FW_SELECT_OCT(8, 0) BO_FW_SELECT_OCT(8, 2)
// four lanes per game: sixteen games per wave-instruction, ten records per lane -- the per-level work that does not grow with the
// records a lane holds (reductions, in-flight masks, row / path bookkeeping, the next request) is shared by twice the games
#define BO_FW_SELECT_QUAD(LCAP, FL, OCC)                                                                               \
    OCC BO_KERNEL void bo_k_fw_select_q4l##LCAP##_##FL(Eng e, FastW f, const float *value, int kind) {                  \
        fw_select_body<4, 10, 1, LCAP, ((FL) & FW_SEL_NT) != 0, ((FL) & FW_SEL_ROOT_IN_REGS) != 0>(e, f, value, kind);  \
    }
BO_FW_SELECT_QUAD(4, 0, ) BO_FW_SELECT_QUAD(4, 2, ) BO_FW_SELECT_QUAD(8, 0, )
BO_FW_SELECT_QUAD(4, 4, BO_FW_OCC(4))   // FW_SEL_DENSE: registers capped for four waves per SIMD = 65 536 games resident at once

// ---- leaf: materialise the position of row r, its legal moves, is_game_over(claim_draw=True), its planes ------------------
BO_KERNEL void bo_k_fw_leaf(Eng e, FastW f, float *nn_in) {
    BO_SHARED FwShared sh;
    BO_SHARED int s_hg[BO_FW_PATH_CAP];
    BO_SHARED unsigned s_flags[BO_FW_PATH_CAP];
    const int L = f.L, g = bo_block() / L, r = bo_block() % L, lane = bo_lane();
    int *ctl = fw_ctl(f, g);
    if (e.phase[g] != PH_RUN || r >= ctl[FWC_NROWS]) return;
    const size_t ro = (size_t)g * L + r;
    const WRec *A = fw_arena(f, g);
    const int slot = ctl[FWC_F(L, FWR_SLOT, r)];
    float *row = nn_in + ro * BO_ROW;
    if (slot == 0) {  // the root: position, legal moves and outcome were prepared with the game stack (root_prepare)
        const DPos P = e.gpos[(size_t)g * e.c.PLY_CAP + e.ply[g]];
        const int n = e.root_nlegal[g];
        for (int j = lane; j < n; j += 64) f.row_moves[ro * BO_MAX_MOVES + j] = e.root_moves[(size_t)g * BO_MAX_MOVES + j];
        if (lane == 0) { f.row_pos[ro] = P; ctl[FWC_F(L, FWR_NLEGAL, r)] = n; ctl[FWC_F(L, FWR_TERM, r)] = 0; bo_atomic_add(&e.stat_evals[g], 1); }
        encode_leaf(e, g, row, P);
        return;
    }
    const DPos P = make_move(fw_head(A, ctl[FWC_F(L, FWR_PLINK, r)])->pos, fw_moves(f, g)[slot]);
    bool chk;
    const int n = bo_movegen(P, sh.moves, &chk);
    const int s = ctl[FWC_F(L, FWR_SIM, r)];
    const int *gpath = f.sim_path + (size_t)g * L * BO_FW_PATH_CAP;  // this game's block: (simulation s, depth k) at FW_PIDX(L, s, k)
    const int d = ctl[FWC_F(L, FWS_PLEN, s)] - 1;  // path[0..d], path[d] = this leaf
    const auto pos_of = [A](int ref) { return fw_head_at(A, ref)->pos; };  // chain refs >= 0: header granule of an ancestor's run
    const int t = terminal_eval_with(e, g, pos_of, P, sh.moves, n, chk, sh.moves2, sh.chain, [&]() {
        // ancestors path[k], k < d, are expanded: their positions head their runs.  Included while every move between
        // them and the leaf is reversible (python-chess pops back to the last irreversible move).
        for (int k = lane; k < d; k += 64) {
            const int hg = fw_first(A[gpath[FW_PIDX(L, s, k)]].link) - BO_FW_HG;
            s_hg[k] = hg;
            s_flags[k] = fw_head_at(A, hg)->pos.flags;
        }
        bo_sync();
        int m = 0;  // largest j in [1, d] whose incoming move was irreversible (j = d is the leaf itself), else 0
        if (P.flags & F_IRREV) m = d;
        else
            for (int j = d - 1; j >= 1; j--)
                if (s_flags[j] & F_IRREV) { m = j; break; }
        int cnt = 0;
        if (m < d) {
            const int lo = m;  // ancestors lo .. d-1 (lo = 0: the root too, then the game's own history follows)
            for (int k = lo + lane; k < d; k += 64) {
                if (k - lo < BO_CHAIN_CAP) { sh.chain.hash[k - lo] = fw_head_at(A, s_hg[k])->pos.khash; sh.chain.ref[k - lo] = s_hg[k]; }
            }
            cnt = d - lo;
        }
        if (m == 0) cnt = chain_collect_history(e, g, sh.chain, cnt);
        bo_sync();
        return cnt;
    });
    if (lane == 0) { f.row_pos[ro] = P; ctl[FWC_F(L, FWR_NLEGAL, r)] = n; ctl[FWC_F(L, FWR_TERM, r)] = t; }
    if (t == 0) {
        for (int j = lane; j < n; j += 64) f.row_moves[ro * BO_MAX_MOVES + j] = sh.moves[j];
        encode_leaf(e, g, row, P);
        if (lane == 0) bo_atomic_add(&e.stat_evals[g], 1);
    }
}

// planes 0..97 into all L rows of game g, phase = RUN; a root kept from the previous search gets its Dirichlet noise here
BO_KERNEL void bo_k_fw_search_begin(Eng e, FastW f, const int *go, float *nn_in) {
    const int g = bo_block(), lane = bo_lane();
    if (!go[g]) return;
    for (int r = 0; r < f.L; r++) encode_static(e, g, nn_in + ((size_t)g * f.L + r) * BO_ROW);
    WRec *A = fw_arena(f, g);
    const int link = A[0].link;
    if (link >= 0 && e.c.use_noise) {
        const int first = fw_first(link), n = e.root_nlegal[g];
        const double *nz = e.noise + (size_t)g * BO_MAX_MOVES;
        for (int j = lane; j < n; j += 64) {
            WRec *c = A + (size_t)first * BO_FW_GR + j;
            const float a = e.c.keep * c->prior;
            c->prior = (float)((double)a + e.c.eps * nz[j]);
        }
    }
    if (lane == 0) { e.phase[g] = PH_RUN; e.sims_done[g] = 0; fw_ctl(f, g)[FWC_NROWS] = 0; fw_ctl(f, g)[FWC_NSTEP] = 0; }
}

// fresh tree (one unexpanded root record in granule 0) for the game slots that were (re)set up
BO_KERNEL void bo_k_fw_reset(Eng e, FastW f, const int *slots) {
    const int g = slots[bo_block()], lane = bo_lane();
    WRec *A = f.arena + fw_arena_off(f, g, 0);
    if (lane < BO_FW_GR) {
        WRec c;
        c.n = lane == 0 ? 0 : -1; c.w = 0.0f; c.prior = lane == 0 ? 1.0f : 0.0f; c.link = FW_UNVISITED;
        A[lane] = c;
    }
    if (lane == 0) {
        int *ctl = fw_ctl(f, g);
        ctl[FWC_CUR] = 0; ctl[FWC_TOP] = 1; ctl[FWC_NROWS] = 0; ctl[FWC_NSTEP] = 0;
        f.played_now[g] = 0;
    }
    (void)e;
}

// Tree reuse: the child reached by the move just played becomes the root; its subtree is copied breadth-first into the
// game's other arena (so the live arena is always compact and in level order), everything else is dropped.
BO_KERNEL void bo_k_fw_reroot(Eng e, FastW f, int reuse) {
    const int g = bo_block(), lane = bo_lane();
    const bo_mv m = (bo_mv)f.played_now[g];
    if (m == 0) return;
    const int c0 = fw_ctl(f, g)[FWC_CUR];
    const WRec *S = f.arena + fw_arena_off(f, g, c0);
    const bo_mv *SM = f.amove + fw_arena_off(f, g, c0);
    WRec *D = f.arena + fw_arena_off(f, g, c0 ^ 1);
    bo_mv *DM = f.amove + fw_arena_off(f, g, c0 ^ 1);
    // the played move among the old root's children
    int child = -1;
    const int rl = S[0].link;
    if (reuse && rl >= 0) {
        const int first = fw_first(rl), nrec = fw_ngran(rl) * BO_FW_GR;
        for (int i0 = 0; i0 < nrec; i0 += 64) {
            const int i = i0 + lane;
            const bool hit = i < nrec && S[(size_t)first * BO_FW_GR + i].n >= 0 && SM[(size_t)first * BO_FW_GR + i] == m;
            const uint64_t b = bo_ballot(hit);
            if (b) { child = first * BO_FW_GR + i0 + bo_lsb64(b); break; }
        }
    }
    const int clink = child >= 0 ? S[child].link : FW_UNVISITED;
    const int cn = (child >= 0 && clink >= 0) ? S[child].n : 0;
    if (lane < BO_FW_GR) {
        WRec c;
        c.n = lane == 0 ? cn : -1; c.w = 0.0f; c.prior = lane == 0 ? 1.0f : 0.0f; c.link = FW_UNVISITED;
        D[lane] = c;
    }
    int top = 1, flags = 0;
    bo_sync();
    if (clink >= 0) {
        // copy the run behind `src_link` (header + records + moves) to D at granule `top`; (new_link) = its link there
        #define BO_FW_COPY_RUN(src_link, new_link)                                                              \
        {                                                                                                       \
            const int _ng = fw_ngran(src_link), _sh = fw_first(src_link) - BO_FW_HG, _tot = (BO_FW_HG + _ng) * BO_FW_GR; \
            for (int _i = lane; _i < _tot; _i += 64) D[(size_t)top * BO_FW_GR + _i] = S[(size_t)_sh * BO_FW_GR + _i];    \
            for (int _i = lane; _i < _ng * BO_FW_GR; _i += 64)                                                  \
                DM[(size_t)(top + BO_FW_HG) * BO_FW_GR + _i] = SM[(size_t)(_sh + BO_FW_HG) * BO_FW_GR + _i];    \
            (new_link) = fw_link(top + BO_FW_HG, _ng);                                                          \
            top += BO_FW_HG + _ng;                                                                              \
        }
        int nl;
        BO_FW_COPY_RUN(clink, nl)
        if (lane == 0) D[0].link = nl;
        bo_sync();
        for (int scan = 1; scan < top;) {  // D's runs in order: [header][records] ...
            const int nrec = fw_gran_for(fw_head_at(D, scan)->nrec) * BO_FW_GR;
            const size_t r0 = (size_t)(scan + BO_FW_HG) * BO_FW_GR;
            for (int i0 = 0; i0 < nrec; i0 += 64) {
                const int i = i0 + lane;
                WRec c;
                c.n = -1; c.link = FW_UNVISITED;
                if (i < nrec) c = D[r0 + i];
                int link = c.link;
                const bool mine = i < nrec && c.n >= 0 && link >= 0;
                uint64_t todo = bo_ballot(mine);
                while (todo) {
                    const int l = bo_lsb64(todo);
                    todo &= todo - 1;
                    const int sl = bo_readlane(link, l);
                    int nl2 = FW_UNVISITED;
                    if (top + BO_FW_HG + fw_ngran(sl) <= f.NG) BO_FW_COPY_RUN(sl, nl2)
                    else flags |= ST_NODE_OVERFLOW;  // cannot happen with the arena sizing of bo_engine_create; the subtree is dropped
                    if (lane == l) link = nl2;
                }
                if (mine) D[r0 + i].link = link;
            }
            bo_sync();
            scan += BO_FW_HG + nrec / BO_FW_GR;
        }
        #undef BO_FW_COPY_RUN
    }
    if (lane == 0) {
        fw_ctl(f, g)[FWC_CUR] = c0 ^ 1; fw_ctl(f, g)[FWC_TOP] = top; f.played_now[g] = 0;
        if (flags) e.status[g] |= flags;
    }
}

// pi over ALL legal root moves = child visits / total; best = first maximum in legal-move order
BO_KERNEL void bo_k_fw_result(Eng e, FastW f) {
    const int g = bo_block(), lane = bo_lane();
    if (g == 0 && lane == 0) e.res_watch[0] = e.watch ? (e.watch[0] | (e.watch_n > 1 && e.watch[1] ? 0x10000 : 0)) : 0;
    if (e.phase[g] != PH_DONE) return;
    const WRec *A = fw_arena(f, g);
    const int link = A[0].link, n = e.root_nlegal[g];
    const int nch = link >= 0 ? n : 0;
    const size_t r0 = link >= 0 ? (size_t)fw_first(link) * BO_FW_GR : 0;
    const bo_mv *mv = e.root_moves + (size_t)g * BO_MAX_MOVES;
    int *ridx = e.res_idx + (size_t)g * BO_RES_CAP;
    float *rval = e.res_val + (size_t)g * BO_RES_CAP;
    int tot = 0, bv = -1, bk = 0x7fffffff;
    for (int i = lane; i < nch; i += 64) {
        const int v = A[r0 + i].n;
        tot += v;
        if (v > bv) { bv = v; bk = i; }
    }
    tot = bo_wave_sum(tot);
    for (int m = 1; m < 64; m <<= 1) {
        const int ov = bo_shfl_xor(bv, m), ok = bo_shfl_xor(bk, m);
        if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
    }
    if (tot > 0) {
        int base = 0;
        for (int i0 = 0; i0 < nch; i0 += 64) {
            const int i = i0 + lane;
            const int v = i < nch ? A[r0 + i].n : 0;
            const uint64_t m = bo_ballot(v > 0);
            if (v > 0) {
                const int o = base + bo_popc64(m & (BIT(lane) - 1));
                ridx[o] = move_to_index(mv[i]);
                rval[o] = (float)((double)v / (double)tot);
            }
            base += bo_popc64(m);
        }
        if (lane == 0) { e.res_n[g] = base; e.res_best_mv[g] = mv[bk]; e.res_best_idx[g] = move_to_index(mv[bk]); e.res_total[g] = tot; }
    } else {
        for (int j = lane; j < n; j += 64) { ridx[j] = move_to_index(mv[j]); rval[j] = (float)(1.0 / (double)n); }
        if (lane == 0) { e.res_n[g] = n; e.res_best_mv[g] = n ? mv[0] : 0; e.res_best_idx[g] = n ? move_to_index(mv[0]) : -1; e.res_total[g] = 0; }
    }
}
