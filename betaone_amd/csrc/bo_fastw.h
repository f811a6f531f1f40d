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
#define FW_UNVISITED (-1)
#define FW_MATE (-2)          // known terminal: the player who moved into the node delivered mate
#define FW_DRAW (-3)
#define FW_SIM_MATE (-2)      // sim_row codes of simulations that ended in a known terminal (no NN row)
#define FW_SIM_DRAW (-3)
enum { FW_SEL_NT = 1, FW_SEL_ROOT_IN_REGS = 2, FW_SEL_DENSE = 4 };  // FastW::sel_flags

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

struct FastW {
    int L, NG;                 // leaves per game per step; granules per arena
    int sel_ut, sel_flags;     // bo_k_fw_select: games per half-wave (1, 2 or 4); FW_SEL_*
    WRec *arena;               // [G][2][NG * GR]   (two arenas per game, side by side: re-rooting compacts from one into the other)
    bo_mv *amove;              // [G][2][NG * GR]   move of each record
    int *cur, *top;            // [G] live arena; granules in use
    int *n_rows, *n_step;      // [G] NN rows / simulations of the step in flight
    int *row_slot, *row_plink, *row_nlegal, *row_term, *row_sim;  // [G][L]  leaf record, link of the run it lives in, ...
    DPos *row_pos;             // [G][L] position of row r's leaf
    bo_mv *row_moves;          // [G][L][256] its legal moves (python-chess order)
    int *sim_row, *sim_plen;   // [G][L] row whose value simulation s backs up (FW_SIM_*: known terminal); its path length
    int *sim_path;             // [G][L][PATH_CAP] record ids root..leaf (record id = granule * GR + index)
    int *played_now;           // [G] move played by the last bo_k_play (0: none); the same array as Eng::played_now
    unsigned long long *stat_gran;        // [G] record granules requested by descents (x 16 * GR = bytes the select path moved)
    unsigned long long *stat_path_nodes;  // [G] path nodes written by the backup (x 16 B algorithmic)
};

BO_DEV size_t fw_arena_off(const FastW &f, int g, int which) { return ((size_t)g * 2 + (size_t)which) * (size_t)f.NG * BO_FW_GR; }
BO_DEV WRec *fw_arena(const FastW &f, int g) { return f.arena + fw_arena_off(f, g, f.cur[g]); }
BO_DEV bo_mv *fw_moves(const FastW &f, int g) { return f.amove + fw_arena_off(f, g, f.cur[g]); }
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
    WRec r;
    r.n = v[0]; r.w = __builtin_bit_cast(float, v[1]); r.prior = __builtin_bit_cast(float, v[2]); r.link = v[3];
    return r;
}
struct fw_nw { int n; float w; };
BO_DEV fw_nw fw_ld_nw(const WRec *p) {
    const fw_i2 v = *reinterpret_cast<const fw_i2 *>(p);
    fw_nw r;
    r.n = v[0]; r.w = __builtin_bit_cast(float, v[1]);
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
    if (e.phase[g] != PH_RUN || r >= f.n_rows[g]) return;
    const size_t ro = (size_t)g * L + r;
    WRec *A = fw_arena(f, g);
    const int slot = f.row_slot[ro], t = f.row_term[ro];
    if (t > 0) {  // found terminal at its first visit: remember it, no children
        if (lane == 0) A[slot].link = t == 1 ? FW_MATE : FW_DRAW;
        return;
    }
    const int n = f.row_nlegal[ro];
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
    int first = f.top[g];
    for (int q = 0; q < r; q++) {
        const int need = fw_row_need(f.row_term[(size_t)g * L + q], f.row_nlegal[(size_t)g * L + q]);
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
// Backup of the step in flight for ONE game, executed by the 32 lanes of its half-wave; lane c owns path depths c and c + 32
// (a record sits at one depth only, so all updates of a record are made by one lane, in simulation order: no cross-lane
// ordering is needed).  CH simulations are fetched at once and chained in registers where their paths share a record, so a
// step's backup costs one memory round trip per CH simulations, not one per simulation.
// Returns the number of known-terminal / found-terminal simulations among them.
#define BO_FW_BK_CH 4
BO_DEV int fw_backup_game(WRec *A, const int *sim_row, const int *sim_plen, const int *sim_path, const int *row_term, const float *vrow,
                          int n_step, int c) {
    int term_sims = 0;
    for (int s0 = 0; s0 < n_step; s0 += BO_FW_BK_CH) {
        float v[BO_FW_BK_CH];
        int plen[BO_FW_BK_CH];
        int maxlen = 0;
        BO_UNROLL
        for (int j = 0; j < BO_FW_BK_CH; j++) {
            const int s = s0 + j;
            const bool ok = s < n_step;
            const int q = ok ? sim_row[s] : FW_SIM_DRAW;
            plen[j] = ok ? sim_plen[s] : 0;
            if (q >= 0) {  // value[] is from the leaf's side to move; the player who moved into the leaf sees -v; a found terminal has its exact value
                const int t = row_term[q];
                v[j] = t > 0 ? (t == 1 ? 1.0f : 0.0f) : -vrow[q];
                term_sims += (ok && t > 0) ? 1 : 0;
            } else {
                v[j] = q == FW_SIM_MATE ? 1.0f : 0.0f;
                term_sims += ok ? 1 : 0;
            }
            maxlen = plen[j] > maxlen ? plen[j] : maxlen;
        }
        for (int k = c; k < maxlen; k += 32) {  // (second pass only for paths deeper than 32)
            int rec[BO_FW_BK_CH];
            bool val[BO_FW_BK_CH];
            fw_nw x[BO_FW_BK_CH];
            BO_UNROLL
            for (int j = 0; j < BO_FW_BK_CH; j++) {
                val[j] = k >= 1 && k < plen[j];  // the root (k = 0) only counts visits: done by the caller
                rec[j] = val[j] ? sim_path[(size_t)(s0 + j) * BO_FW_PATH_CAP + k] : 0;
            }
            BO_UNROLL
            for (int j = 0; j < BO_FW_BK_CH; j++) {
                x[j].n = 0; x[j].w = 0.0f;
                if (val[j]) x[j] = fw_ld_nw(A + rec[j]);
            }
            BO_UNROLL
            for (int j = 0; j < BO_FW_BK_CH; j++) {
                if (!val[j]) continue;
                BO_UNROLL
                for (int i = 0; i < j; i++)
                    if (val[i] && rec[i] == rec[j]) x[j] = x[i];  // (the latest earlier simulation through the same record wins)
                const float sgn = ((plen[j] - 1 - k) & 1) ? -v[j] : v[j];
                x[j].n = x[j].n + 1;
                x[j].w = x[j].w + sgn;
            }
            BO_UNROLL
            for (int j = 0; j < BO_FW_BK_CH; j++)
                if (val[j]) fw_st_nw(A + rec[j], x[j].n, x[j].w);
        }
    }
    return term_sims;
}

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

// NT: runs below the root are requested non-temporally (read once per launch); the root's run, read by every descent of the
// step, takes the cached path.  ROOTC: the first 64 records of the root's run stay in registers for the whole launch.
template <int UT, int LCAP, bool NT, bool ROOTC>
BO_DEV void fw_select_body(const Eng &e, const FastW &f, const float *value, int kind) {
    typedef typename FwMask<LCAP>::T mask_t;
    BO_SHARED unsigned char s_idx[2 * UT][LCAP][BO_FW_PATH_CAP];  // child index chosen at depth d by descent s of the step
    BO_SHARED int s_rowslot[2 * UT][LCAP];                         // leaf record of the step's NN rows
    BO_SHARED int s_path[2 * UT][BO_FW_PATH_CAP];                  // record ids of the descent in progress
    const int lane = bo_lane(), half = lane >> 5, c = lane & 31, hb = half << 5;
    const int L = f.L, S = e.c.S;
    const float cpuct = e.c.cpuct;
    // the 2 * UT games of a workgroup are consecutive: their arenas are addressed as one uniform base + a 32-bit offset
    const int g0 = bo_block() * 2 * UT;
    WRec *const base = f.arena + fw_arena_off(f, g0, 0);
    unsigned aoff[UT];  // records from `base` to the game's live arena
    int sims[UT], n_rows[UT], n_step[UT], root_n[UT], root_link[UT];
    int link[UT], pn[UT], d[UT], cur[UT], lastlink[UT];
    int lg[UT], kids[UT];  // levels | record granules requested << 12; children scanned (this lane's share)
    mask_t M[UT];
    bool on[UT], busy[UT], done[UT];
    WRec rr0[ROOTC ? UT : 1], rr1[ROOTC ? UT : 1];
#define FW_G(u) (g0 + half * UT + (u))
#define FW_A(u) (base + aoff[u])

    // ---- 1. the previous step's rows have been applied: back their values up ---------------------------------------------
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        on[u] = FW_G(u) < e.c.G && e.phase[FW_G(u)] == PH_RUN;
        const int gg = on[u] ? FW_G(u) : g0;
        aoff[u] = (unsigned)(((size_t)(gg - g0) * 2 + (size_t)f.cur[gg]) * (size_t)f.NG * BO_FW_GR);
        sims[u] = e.sims_done[gg]; n_rows[u] = f.n_rows[gg]; n_step[u] = f.n_step[gg];
        lg[u] = kids[u] = 0;
        busy[u] = done[u] = false;
        link[u] = -1; pn[u] = 1; d[u] = 1; cur[u] = 0; lastlink[u] = -1; M[u] = 0;
        root_n[u] = 0; root_link[u] = -1;
        if (on[u] && n_rows[u] > 0 && kind == POLICY_NONE) on[u] = false;  // (rows waiting for an evaluation that has not been made)
        if (on[u]) root_n[u] = FW_A(u)[0].n;
        if (on[u] && n_rows[u] > 0) {
            const size_t go = (size_t)gg * L;
            int top = f.top[gg];
            for (int q = 0; q < n_rows[u]; q++) {
                const int need = fw_row_need(f.row_term[go + q], f.row_nlegal[go + q]);
                if (top + need <= f.NG) top += need;  // (bo_k_fw_apply refused the runs that do not fit, in the same order)
            }
            if (f.row_slot[go] == 0 && n_step[u] == 0) root_n[u] = 1;  // the root's own evaluation counts as its first visit
            const int ts = fw_backup_game(FW_A(u), f.sim_row + go, f.sim_plen + go, f.sim_path + go * BO_FW_PATH_CAP, f.row_term + go, value + go,
                                          n_step[u], c);
            root_n[u] += n_step[u];
            sims[u] += n_step[u];
            n_rows[u] = n_step[u] = 0;
            if (c == 0) { f.top[gg] = top; e.stat_term_sims[gg] += ts; }
        }
    }
    bo_sync();  // the backup's stores are complete before any descent reads the tree

    // ---- 2. up to L descents per game; the tree is read-only from here on ------------------------------------------------
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        if (!on[u]) continue;
        const int gg = FW_G(u);
        root_link[u] = FW_A(u)[0].link;
        if (e.root_term[gg] != 0 || (root_link[u] >= 0 && sims[u] >= S)) {
            done[u] = true;
        } else if (root_link[u] < 0) {  // root not expanded yet: its evaluation is row 0 (no simulation attached)
            if (c == 0) {
                f.row_slot[(size_t)gg * L] = 0; f.row_plink[(size_t)gg * L] = -1; f.row_sim[(size_t)gg * L] = -1;
                s_rowslot[half * UT + u][0] = 0;
            }
            n_rows[u] = 1;
        } else {
            busy[u] = true;
            link[u] = root_link[u]; pn[u] = root_n[u] + 1;
            if (c == 0) s_path[half * UT + u][0] = 0;
            if (ROOTC) {
                const int nrec = fw_ngran(root_link[u]) * BO_FW_GR;
                const WRec *R = FW_A(u) + (size_t)fw_first(root_link[u]) * BO_FW_GR;
                WRec pad; pad.n = -1; pad.w = 0.0f; pad.prior = 0.0f; pad.link = FW_UNVISITED;
                rr0[u] = c < nrec ? fw_ld<false>(R + c) : pad;
                rr1[u] = c + 32 < nrec ? fw_ld<false>(R + c + 32) : pad;
                lg[u] += fw_ngran(root_link[u]) << 12;
            }
        }
    }
    bool any = false;
    BO_UNROLL
    for (int u = 0; u < UT; u++) any = any || busy[u];
    while (bo_ballot(any) != 0) {
        WRec r0[UT], r1[UT];
        // every game's run is requested before any is consumed
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            WRec pad; pad.n = -1; pad.w = 0.0f; pad.prior = 0.0f; pad.link = FW_UNVISITED;
            r0[u] = pad; r1[u] = pad;
            if (busy[u]) {
                const int nrec = fw_ngran(link[u]) * BO_FW_GR;
                const WRec *R = FW_A(u) + (size_t)fw_first(link[u]) * BO_FW_GR;
                if (ROOTC && d[u] == 1) { r0[u] = rr0[u]; r1[u] = rr1[u]; }
                else if (NT && d[u] > 1) {
                    if (c < nrec) r0[u] = fw_ld<true>(R + c);
                    if (c + 32 < nrec) r1[u] = fw_ld<true>(R + c + 32);
                } else {
                    if (c < nrec) r0[u] = fw_ld<false>(R + c);
                    if (c + 32 < nrec) r1[u] = fw_ld<false>(R + c + 32);
                }
            }
        }
        any = false;
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            const int slot = half * UT + u;
            const bool lv = busy[u];
            const int ngran = fw_ngran(link[u]), nrec = ngran * BO_FW_GR, first = fw_first(link[u]);
            const int dd = d[u] < BO_FW_PATH_CAP ? d[u] : BO_FW_PATH_CAP - 1;
            const float sq = sqrtf((float)pn[u]);
            float best = -__builtin_inff();
            int bi = 0x7fffffff, bne = 0, bl = FW_UNVISITED;
            // descents of this step in flight through this lane's candidates (those that share the whole path so far)
            int cnt0 = 0, cnt1 = 0;
            if (lv)
                for (mask_t m = M[u]; m; m &= m - 1) {
                    const int b = s_idx[slot][fw_ctz(m)][dd];
                    cnt0 += b == c ? 1 : 0;
                    cnt1 += b == c + 32 ? 1 : 0;
                }
            // one candidate record: its statistics with the descents in flight through it, its PUCT score
#define BO_FW_CAND(rec, idx, cnt)                                                                               \
            if (lv && (rec).n >= 0) {                                                                           \
                const int ne = (rec).n + (cnt);                                                                 \
                const float we = (rec).w - (float)(cnt);                                                        \
                const float t1 = cpuct * (rec).prior;                                                           \
                const float t2 = t1 * sq;                                                                       \
                const float uu = t2 / (float)(1 + ne);                                                          \
                const float qv = ne > 0 ? we / (float)ne : 0.0f;                                                \
                const float sc = qv + uu;                                                                       \
                kids[u]++;                                                                                      \
                if (sc > best) { best = sc; bi = (idx); bne = ne; bl = (rec).link; }                            \
            }
            BO_FW_CAND(r0[u], c, cnt0)
            BO_FW_CAND(r1[u], c + 32, cnt1)
            if (lv && nrec > 64) {  // a run of more than 64 records (rare: > 64 legal moves)
                const WRec *R = FW_A(u) + (size_t)first * BO_FW_GR;
                for (int i = 64 + c; i < nrec; i += 32) {
                    const WRec rx = fw_ld<false>(R + i);
                    int cntx = 0;
                    for (mask_t m = M[u]; m; m &= m - 1) cntx += s_idx[slot][fw_ctz(m)][dd] == i ? 1 : 0;
                    BO_FW_CAND(rx, i, cntx)
                }
            }
#undef BO_FW_CAND
            BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 0)), BO_ROW_XCHG(bi, 0))
            BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 1)), BO_ROW_XCHG(bi, 1))
            BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 2)), BO_ROW_XCHG(bi, 2))
            BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 3)), BO_ROW_XCHG(bi, 3))
            BO_FW_ARGMAX(bo_shfl_xor_f(best, 16), bo_shfl_xor(bi, 16))
            bool nan_all = false;
            if (bi >= nrec) {  // every score was NaN: take the first child (it exists: a run is never empty)
                nan_all = true;
                bi = 0;
                if (c == 0) { bne = r0[u].n; bl = r0[u].link; }  // (in-flight visits are not added here: the search is broken anyway)
            }
            // the winner's statistics come from the lane that scored it (index bi belongs to lane bi & 31, whose own best it is)
            const int src = hb + (bi & 31);
            const int w_ne = bo_shfl(bne, src);
            const int w_l = bo_shfl(bl, src);
            if (lv) {
                if (nan_all && c == 0) bo_atomic_or(&e.status[FW_G(u)], ST_NAN_SCORE);
                for (mask_t m = M[u]; m; m &= m - 1) {  // earlier descents that went elsewhere no longer share the path
                    const int b = fw_ctz(m);
                    if (s_idx[slot][b][dd] != (unsigned char)bi) M[u] &= ~((mask_t)1 << b);
                }
                lastlink[u] = link[u];
                cur[u] = first * BO_FW_GR + bi;
                if (c == 0) { s_idx[slot][n_step[u]][dd] = (unsigned char)bi; s_path[slot][dd] = cur[u]; }
                lg[u] += 1 + ((ROOTC && d[u] == 1) ? 0 : (ngran << 12));
                pn[u] = w_ne + 1;
                d[u]++;
                link[u] = w_l;
            }
            // ---- end of a descent? -----------------------------------------------------------------------------------
            const bool ended = lv && (link[u] < 0 || d[u] >= BO_FW_PATH_CAP);
            if (ended && link[u] >= 0) {  // path buffer full: the visit counts as a draw
                link[u] = FW_DRAW;
                if (c == 0) bo_atomic_or(&e.status[FW_G(u)], ST_DEPTH_OVERFLOW);
            }
            // a leaf another descent of this step already selected shares that descent's row
            const bool fresh = ended && link[u] == FW_UNVISITED;
            const uint64_t h0 = bo_ballot(fresh && c < n_rows[u] && s_rowslot[slot][c < LCAP ? c : 0] == cur[u]);
            const uint64_t h1 = LCAP > 32 ? bo_ballot(fresh && c + 32 < n_rows[u] && s_rowslot[slot][c + 32 < LCAP ? c + 32 : 0] == cur[u]) : 0ull;
            if (ended) {
                const int gg = FW_G(u);
                const size_t go = (size_t)gg * L;
                const unsigned m0 = (unsigned)(h0 >> hb), m1 = (unsigned)(h1 >> hb);
                int q;
                if (link[u] == FW_MATE) q = FW_SIM_MATE;
                else if (link[u] == FW_DRAW) q = FW_SIM_DRAW;
                else if (m0) q = __builtin_ctz(m0);
                else if (m1) q = 32 + __builtin_ctz(m1);
                else {  // becomes NN row n_rows
                    q = n_rows[u];
                    if (c == 0) {
                        s_rowslot[slot][q] = cur[u];
                        f.row_slot[go + q] = cur[u]; f.row_plink[go + q] = lastlink[u]; f.row_sim[go + q] = n_step[u];
                    }
                    n_rows[u]++;
                }
                const int s = n_step[u];
                if (c == 0) { f.sim_row[go + s] = q; f.sim_plen[go + s] = d[u]; }
                int *path = f.sim_path + (go + s) * BO_FW_PATH_CAP;
                if (c < d[u]) path[c] = s_path[slot][c];
                if (c + 32 < d[u]) path[c + 32] = s_path[slot][c + 32];
                n_step[u] = s + 1;
                if (n_step[u] < L && sims[u] + n_step[u] < S) {  // the game's next descent starts at the root
                    link[u] = root_link[u]; pn[u] = root_n[u] + n_step[u] + 1; d[u] = 1; cur[u] = 0;
                    M[u] = (mask_t)(((mask_t)1 << n_step[u]) - 1);
                } else {
                    busy[u] = false;
                }
            }
            any = any || busy[u];
        }
    }

    // ---- 3. a step of known-terminal hits only needs no evaluation: account for it now -----------------------------------
    int pnodes[UT];
    bool allterm = false;
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        pnodes[u] = (lg[u] & 0xFFF) + n_step[u];  // a path holds the root and one node per level
        allterm = allterm || (on[u] && n_rows[u] == 0 && n_step[u] > 0);
    }
    if (bo_ballot(allterm) != 0) {
        bo_sync();  // the paths written above are complete
        BO_UNROLL
        for (int u = 0; u < UT; u++) {
            if (!(on[u] && n_rows[u] == 0 && n_step[u] > 0)) continue;
            const size_t go = (size_t)FW_G(u) * L;
            const int ts = fw_backup_game(FW_A(u), f.sim_row + go, f.sim_plen + go, f.sim_path + go * BO_FW_PATH_CAP, f.row_term + go, value + go,
                                          n_step[u], c);
            root_n[u] += n_step[u];
            sims[u] += n_step[u];
            n_step[u] = 0;
            if (sims[u] >= S) done[u] = true;
            if (c == 0) e.stat_term_sims[FW_G(u)] += ts;
        }
    }
    BO_UNROLL
    for (int u = 0; u < UT; u++) {
        int k = kids[u];
        for (int m = 1; m < 32; m <<= 1) k += bo_shfl_xor(k, m);
        if (on[u] && c == 0) {
            const int gg = FW_G(u);
            FW_A(u)[0].n = root_n[u];
            e.sims_done[gg] = sims[u]; e.phase[gg] = done[u] ? PH_DONE : PH_RUN;
            f.n_rows[gg] = n_rows[u]; f.n_step[gg] = n_step[u];
            e.req_node[gg] = n_rows[u] > 0 ? s_rowslot[half * UT + u][0] : -1;
            if (lg[u]) {
                e.stat_levels[gg] += lg[u] & 0xFFF; e.stat_children_scanned[gg] += k;
                f.stat_gran[gg] += (unsigned long long)(lg[u] >> 12); f.stat_path_nodes[gg] += (unsigned long long)pnodes[u];
            }
        }
    }
#undef FW_G
#undef FW_A
}
#undef BO_FW_ARGMAX

// One instantiation per (games per half-wave, leaves-per-step capacity of the LDS path bytes, FW_SEL_* flags).  FW_SEL_DENSE:
// the register allocation is capped so that one more wave fits per SIMD (more runs in flight per CU, some spills).
#if defined(BO_WAVE_EMU)
#define BO_FW_OCC(w)
#else
#define BO_FW_OCC(w) __attribute__((amdgpu_waves_per_eu(w)))
#endif
#define BO_FW_SELECT_KERNEL(UT, LCAP, FL, OCC)                                                                         \
    OCC BO_KERNEL void bo_k_fw_select_u##UT##l##LCAP##_##FL(Eng e, FastW f, const float *value, int kind) {            \
        fw_select_body<UT, LCAP, ((FL) & FW_SEL_NT) != 0, ((FL) & FW_SEL_ROOT_IN_REGS) != 0>(e, f, value, kind);        \
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

// ---- leaf: materialise the position of row r, its legal moves, is_game_over(claim_draw=True), its planes ------------------
BO_KERNEL void bo_k_fw_leaf(Eng e, FastW f, float *nn_in) {
    BO_SHARED FwShared sh;
    BO_SHARED int s_hg[BO_FW_PATH_CAP];
    BO_SHARED unsigned s_flags[BO_FW_PATH_CAP];
    const int L = f.L, g = bo_block() / L, r = bo_block() % L, lane = bo_lane();
    if (e.phase[g] != PH_RUN || r >= f.n_rows[g]) return;
    const size_t ro = (size_t)g * L + r;
    const WRec *A = fw_arena(f, g);
    const int slot = f.row_slot[ro];
    float *row = nn_in + ro * BO_ROW;
    if (slot == 0) {  // the root: position, legal moves and outcome were prepared with the game stack (root_prepare)
        const DPos P = e.gpos[(size_t)g * e.c.PLY_CAP + e.ply[g]];
        const int n = e.root_nlegal[g];
        for (int j = lane; j < n; j += 64) f.row_moves[ro * BO_MAX_MOVES + j] = e.root_moves[(size_t)g * BO_MAX_MOVES + j];
        if (lane == 0) { f.row_pos[ro] = P; f.row_nlegal[ro] = n; f.row_term[ro] = 0; bo_atomic_add(&e.stat_evals[g], 1); }
        encode_leaf(e, g, row, P);
        return;
    }
    const DPos P = make_move(fw_head(A, f.row_plink[ro])->pos, fw_moves(f, g)[slot]);
    bool chk;
    const int n = bo_movegen(P, sh.moves, &chk);
    const int s = f.row_sim[ro];
    const int *path = f.sim_path + ((size_t)g * L + s) * BO_FW_PATH_CAP;
    const int d = f.sim_plen[(size_t)g * L + s] - 1;  // path[0..d], path[d] = this leaf
    const auto pos_of = [A](int ref) { return fw_head_at(A, ref)->pos; };  // chain refs >= 0: header granule of an ancestor's run
    const int t = terminal_eval_with(e, g, pos_of, P, sh.moves, n, chk, sh.moves2, sh.chain, [&]() {
        // ancestors path[k], k < d, are expanded: their positions head their runs.  Included while every move between
        // them and the leaf is reversible (python-chess pops back to the last irreversible move).
        for (int k = lane; k < d; k += 64) {
            const int hg = fw_first(A[path[k]].link) - BO_FW_HG;
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
    if (lane == 0) { f.row_pos[ro] = P; f.row_nlegal[ro] = n; f.row_term[ro] = t; }
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
    if (lane == 0) { e.phase[g] = PH_RUN; e.sims_done[g] = 0; f.n_rows[g] = 0; f.n_step[g] = 0; }
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
    if (lane == 0) { f.cur[g] = 0; f.top[g] = 1; f.n_rows[g] = 0; f.n_step[g] = 0; f.played_now[g] = 0; }
    (void)e;
}

// Tree reuse: the child reached by the move just played becomes the root; its subtree is copied breadth-first into the
// game's other arena (so the live arena is always compact and in level order), everything else is dropped.
BO_KERNEL void bo_k_fw_reroot(Eng e, FastW f, int reuse) {
    const int g = bo_block(), lane = bo_lane();
    const bo_mv m = (bo_mv)f.played_now[g];
    if (m == 0) return;
    const int c0 = f.cur[g];
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
        f.cur[g] = c0 ^ 1; f.top[g] = top; f.played_now[g] = 0;
        if (flags) e.status[g] |= flags;
    }
}

// pi over ALL legal root moves = child visits / total; best = first maximum in legal-move order
BO_KERNEL void bo_k_fw_result(Eng e, FastW f) {
    const int g = bo_block(), lane = bo_lane();
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
