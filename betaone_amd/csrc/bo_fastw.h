// betaone_amd/csrc/bo_fastw.h -- FAST search mode.  NOT the reference's semantics.
//
// The reference's search has no virtual loss, so every NN batch holds one position repeated up to 96 times and the root
// never gets more than two children (mcts.py:186,203,210-254).  This mode keeps the reference's
// interfaces (game stack, legal-move order, draw rules, input planes, (state, pi, z) records) but runs a conventional
// batched AlphaZero-style search, clearly labelled as diverging from the reference:
//   * L leaves per game per step, selected one after another with a VIRTUAL LOSS (n += 1, W -= 1 on the path);
//     NN batch = G x L rows, row = g*L + r; two descents that end in the same unexpanded leaf share its row;
//   * full-width expansion: every legal move becomes a child, prior = policy mass renormalised over the legal moves;
//     Dirichlet noise on all root priors;
//   * W of a node is the value sum seen by the player who moved INTO it, so PUCT needs no negation:
//       score = W/n + cpuct * P * sqrt(N_parent) / (1 + n);
//   * TREE REUSE: after a move the played child's subtree becomes the next search's tree (the reference rebuilds the
//     tree from scratch every move, mcts.py:176).
//
// Layout for bandwidth (the form of the select kernel that SURVEY.md section 8d prices against the HBM roofline):
// the children of a node are ONE run of 512-byte, 512-byte-aligned CHILD BLOCKS of 32 16-byte records
//     { int32 n; float w; float prior; int32 link }
// (link: the child's own run = first block | (blocks - 1) << 24, or a negative state code), so a level of a descent is one
// coalesced 512-B request per block -- 12 B/child of statistics plus the 4-B link that shares the cache lines anyway -- and
// the virtual loss / the backup rewrite 8 bytes of the SAME records.  Everything that is not on the select path lives
// elsewhere: the move of a record (2 B, parallel array), the position of an expanded node (80 B per run).  A tree is a
// per-game arena of blocks with a bump allocator; re-rooting copies the kept subtree breadth-first into the game's second
// arena (compaction and garbage collection in one pass).
//
// One step = three launches inside the captured graph:
//   bo_k_fw_apply   one wave per (game, row): priors of the evaluated leaf -> a new run of child blocks
//   bo_k_fw_select  one wave per game: backup of the previous step's simulations (virtual loss removed in the same pass),
//                   then L descents with virtual loss                                   <- the select + backup kernel
//   bo_k_fw_leaf    one wave per (game, row): make-move, legal moves, is_game_over(claim_draw=True), planes 98..119
// Arithmetic is plain binary32 in a fixed order; tests/fast_reference.py restates it in NumPy and whole trees are compared
// bit for bit (there is no reference implementation of this mode to compare with).
#pragma once
#include "bo_tree.h"

#define BO_FW_C 32            // records per child block
#define BO_FW_PATH_CAP 64     // deepest path of one descent
#define BO_FW_LINK_MASK 0xFFFFFF
#define FW_UNVISITED (-1)
#define FW_MATE (-2)          // known terminal: the player who moved into the node delivered mate
#define FW_DRAW (-3)
#define FW_PENDING(r) (-16 - (r))   // selected in this step, its evaluation is NN row r

struct WRec {
    int n;        // visits (incl. virtual ones); -1 = padding of a run's last block, never selected
    float w;      // value sum from the point of view of the player who moved into the node (virtual loss: -1 per visit in flight)
    float prior;
    int link;
};

struct FastW {
    int L, NB;                 // leaves per game per step; blocks per arena
    WRec *arena[2];            // [G][NB][32]   (two arenas per game: re-rooting compacts from one into the other)
    bo_mv *amove[2];           // [G][NB*32]    move of each record
    DPos *bpos[2];             // [G][NB]       position of the node whose children start at this block
    int *cur, *top;            // [G] live arena; blocks in use
    int *n_rows, *n_step;      // [G] NN rows / simulations of the step in flight
    int *row_slot, *row_prun, *row_nlegal, *row_term, *row_sim;  // [G][L]
    DPos *row_pos;             // [G][L] position of row r's leaf
    bo_mv *row_moves;          // [G][L][256] its legal moves (python-chess order)
    int *sim_row, *sim_plen;   // [G][L] row whose value simulation s backs up (-1: known terminal, already backed up)
    int *sim_path;             // [G][L][PATH_CAP] record ids root..leaf (record id = block*32 + index)
    int *played_now;           // [G] move played by the last bo_k_play (0: none); the same array as Eng::played_now
    unsigned long long *stat_blocks;  // [G] child blocks read by descents (x 512 B = bytes the select path moved)
    unsigned long long *stat_path_nodes;  // [G] path nodes written by virtual loss + backup (x 16 B algorithmic)
};

BO_DEV WRec *fw_arena(const FastW &f, int g) { return f.arena[f.cur[g]] + (size_t)g * f.NB * BO_FW_C; }
BO_DEV bo_mv *fw_moves(const FastW &f, int g) { return f.amove[f.cur[g]] + (size_t)g * f.NB * BO_FW_C; }
BO_DEV DPos *fw_bpos(const FastW &f, int g) { return f.bpos[f.cur[g]] + (size_t)g * f.NB; }
BO_DEV int fw_nblk(int link) { return ((link >> 24) & 7) + 1; }

// lane-strided partial sums + butterfly: the summation order tests/fast_reference.py mirrors
BO_DEV float fw_sum(const float *v, int n) {
    float a = 0.0f;
    for (int j = bo_lane(); j < n; j += 64) a = a + v[j];
    return bo_wave_sum_f(a);
}

// remove the virtual loss of one simulation and add its value: lane k handles path node k (the parent chain is written in parallel)
BO_DEV void fw_backup(WRec *A, const int *path, int plen, float v_leaf_mover) {
    for (int k = bo_lane(); k < plen; k += 64) {
        if (k == 0) continue;  // root: only its visit count matters (incremented at selection)
        const float s = ((plen - 1 - k) & 1) ? -v_leaf_mover : v_leaf_mover;
        WRec *r = A + path[k];
        r->w = (r->w + 1.0f) + s;
    }
}

struct FwShared {
    bo_mv moves[BO_MAX_MOVES];
    bo_mv moves2[BO_MAX_MOVES];
    float pv[BO_MAX_MOVES];
    ChainBuf chain;
};

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
    for (int q = 0; q < r; q++)
        if (f.row_term[(size_t)g * L + q] == 0) {
            const int nb = (f.row_nlegal[(size_t)g * L + q] + BO_FW_C - 1) / BO_FW_C;
            if (first + nb <= f.NB) first += nb;  // (a run that does not fit is refused; bo_k_fw_select advances `top` by the same rule)
        }
    const int nblk = (n + BO_FW_C - 1) / BO_FW_C;
    if (first + nblk > f.NB) {  // arena full: the leaf stays unexpanded (its value is still backed up)
        if (lane == 0) { A[slot].link = FW_UNVISITED; bo_atomic_or(&e.status[g], ST_NODE_OVERFLOW); }
        return;
    }
    bo_mv *M = fw_moves(f, g);
    for (int i = lane; i < nblk * BO_FW_C; i += 64) {
        WRec c;
        c.n = i < n ? 0 : -1; c.w = 0.0f; c.prior = i < n ? pv[i] : 0.0f; c.link = FW_UNVISITED;
        A[(size_t)first * BO_FW_C + i] = c;
        M[(size_t)first * BO_FW_C + i] = i < n ? mv[i] : (bo_mv)0;
    }
    if (lane == 0) {
        fw_bpos(f, g)[first] = f.row_pos[ro];
        A[slot].link = first | ((nblk - 1) << 24);
    }
}

// ---- select + backup ----------------------------------------------------------------------------------------------------
BO_KERNEL void bo_k_fw_select(Eng e, FastW f, const float *value, int kind) {
    const int g = bo_block(), lane = bo_lane(), L = f.L;
    if (e.phase[g] != PH_RUN) return;
    WRec *A = fw_arena(f, g);
    int sims = e.sims_done[g], n_rows = f.n_rows[g], n_step = f.n_step[g], flags = 0;
    int *row_slot = f.row_slot + (size_t)g * L, *row_prun = f.row_prun + (size_t)g * L, *row_sim = f.row_sim + (size_t)g * L;
    const int *row_term = f.row_term + (size_t)g * L, *row_nl = f.row_nlegal + (size_t)g * L;
    int *sim_row = f.sim_row + (size_t)g * L, *sim_plen = f.sim_plen + (size_t)g * L;
    int *sim_path = f.sim_path + (size_t)g * L * BO_FW_PATH_CAP;

    // ---- 1. the previous step's rows have been applied: back their values up, release the virtual loss -----------------
    if (n_rows > 0) {
        if (kind == POLICY_NONE) return;
        int top = f.top[g], term_sims = 0;
        for (int q = 0; q < n_rows; q++)
            if (row_term[q] == 0) {
                const int nb = (row_nl[q] + BO_FW_C - 1) / BO_FW_C;
                if (top + nb <= f.NB) top += nb;  // (bo_k_fw_apply refused the runs that do not fit, in the same order)
            }
        if (row_slot[0] == 0 && n_step == 0 && lane == 0) A[0].n = 1;  // the root's own evaluation counts as its first visit
        for (int s = 0; s < n_step; s++) {
            const int q = sim_row[s];
            if (q < 0) continue;
            const int t = row_term[q];
            // value[] is from the leaf's side to move; the player who moved into the leaf sees -v; a terminal leaf has its exact value
            const float v = t > 0 ? (t == 1 ? 1.0f : 0.0f) : -value[(size_t)g * L + q];
            term_sims += t > 0 ? 1 : 0;
            fw_backup(A, sim_path + (size_t)s * BO_FW_PATH_CAP, sim_plen[s], v);
            bo_sync();
        }
        sims += n_step;
        n_rows = n_step = 0;
        if (lane == 0) { f.top[g] = top; e.stat_term_sims[g] += term_sims; }
        bo_sync();
    }

    // ---- 2. up to L descents with virtual loss -------------------------------------------------------------------------
    int phase = PH_RUN;
    const int root_link = A[0].link;
    if (e.root_term[g] != 0 || (root_link >= 0 && sims >= e.c.S)) {
        phase = PH_DONE;
    } else if (root_link < 0) {  // root not expanded yet: its evaluation is row 0 (no simulation attached)
        if (lane == 0) { row_slot[0] = 0; row_prun[0] = -1; row_sim[0] = -1; }
        n_rows = 1;
    } else {
        int levels = 0, blocks = 0, kids = 0, pnodes = 0;
        while (n_step < L && sims + n_step < e.c.S) {
            int *path = sim_path + (size_t)n_step * BO_FW_PATH_CAP;
            int cur = 0, d = 1, link = root_link, prun = -1;
            int pn = A[0].n + 1;  // visits of the node being expanded, this simulation included
            if (lane == 0) { path[0] = 0; A[0].n = pn; }
            while (link >= 0 && d < BO_FW_PATH_CAP) {
                const int first = link & BO_FW_LINK_MASK, nblk = fw_nblk(link);
                const float sq = sqrtf((float)pn);
                float best = -__builtin_inff();
                int bi = 0x7fffffff, bn = 0, bl = FW_UNVISITED;
                float bw = 0.0f;
                for (int i0 = 0; i0 < nblk * BO_FW_C; i0 += 64) {  // two child blocks per pass, one 16-byte record per lane
                    const int i = i0 + lane;
                    if (i < nblk * BO_FW_C) {
                        const WRec c = A[(size_t)first * BO_FW_C + i];
                        if (c.n >= 0) {
                            kids++;
                            const float t1 = e.c.cpuct * c.prior;
                            const float t2 = t1 * sq;
                            const float u = t2 / (float)(1 + c.n);
                            const float qv = c.n > 0 ? c.w / (float)c.n : 0.0f;
                            const float sc = qv + u;
                            if (sc > best) { best = sc; bi = i; bn = c.n; bw = c.w; bl = c.link; }
                        }
                    }
                }
                // first maximum in child order: (score, index) through four DPP row rounds and two cross-row exchanges; the
                // winner's statistics then come from its lane (index & 63) by v_readlane
#define BO_FW_ARGMAX(os_expr, oi_expr)                                                      \
                {                                                                           \
                    const float os = (os_expr);                                             \
                    const int oi = (oi_expr);                                               \
                    if (os > best || (os == best && oi < bi)) { best = os; bi = oi; }       \
                }
                BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 0)), BO_ROW_XCHG(bi, 0))
                BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 1)), BO_ROW_XCHG(bi, 1))
                BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 2)), BO_ROW_XCHG(bi, 2))
                BO_FW_ARGMAX(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, best), 3)), BO_ROW_XCHG(bi, 3))
                BO_FW_ARGMAX(bo_shfl_xor_f(best, 16), bo_shfl_xor(bi, 16))
                BO_FW_ARGMAX(bo_shfl_xor_f(best, 32), bo_shfl_xor(bi, 32))
#undef BO_FW_ARGMAX
                bi = bo_uniform(bi);
                if (bi < nblk * BO_FW_C) {
                    const int src = bi & 63;
                    bn = bo_readlane(bn, src);
                    bw = __builtin_bit_cast(float, bo_readlane(__builtin_bit_cast(int, bw), src));
                    bl = bo_readlane(bl, src);
                }
                if (bi >= nblk * BO_FW_C) {  // every score was NaN: take the first child (it exists: a run is never empty)
                    const WRec c = A[(size_t)first * BO_FW_C];
                    bi = 0; bn = c.n; bw = c.w; bl = c.link;
                    flags |= ST_NAN_SCORE;
                }
                levels++;
                blocks += nblk;
                prun = first;
                cur = first * BO_FW_C + bi;
                pn = bn + 1;
                if (lane == 0) {
                    path[d] = cur;
                    A[cur].n = pn;           // virtual loss
                    A[cur].w = bw - 1.0f;
                }
                d++;
                link = bl;
                bo_sync();
            }
            if (link >= 0) { link = FW_DRAW; flags |= ST_DEPTH_OVERFLOW; }  // path buffer full: the visit counts as a draw
            int q = -1;
            if (link == FW_MATE || link == FW_DRAW) {  // known terminal: exact value now
                fw_backup(A, path, d, link == FW_MATE ? 1.0f : 0.0f);
                if (lane == 0) e.stat_term_sims[g] += 1;
            } else if (link == FW_UNVISITED) {  // becomes NN row n_rows
                q = n_rows;
                if (lane == 0) { A[cur].link = FW_PENDING(q); row_slot[q] = cur; row_prun[q] = prun; row_sim[q] = n_step; }
                n_rows++;
            } else {
                q = -16 - link;  // already selected in this step: share its row
            }
            if (lane == 0) { sim_row[n_step] = q; sim_plen[n_step] = d; }
            pnodes += d;
            n_step++;
            bo_sync();
        }
        kids = bo_wave_sum(kids);
        if (lane == 0) {
            e.stat_levels[g] += levels; e.stat_children_scanned[g] += kids;
            f.stat_blocks[g] += (unsigned long long)blocks; f.stat_path_nodes[g] += (unsigned long long)pnodes;
        }
        if (n_rows == 0) {  // only known-terminal hits in this step: account for them now
            sims += n_step;
            n_step = 0;
            if (sims >= e.c.S) phase = PH_DONE;
        }
    }
    if (lane == 0) {
        e.sims_done[g] = sims; e.phase[g] = phase;
        f.n_rows[g] = n_rows; f.n_step[g] = n_step;
        e.req_node[g] = n_rows > 0 ? row_slot[0] : -1;
        if (flags) e.status[g] |= flags;
    }
}

// ---- leaf: materialise the position of row r, its legal moves, is_game_over(claim_draw=True), its planes ------------------
BO_KERNEL void bo_k_fw_leaf(Eng e, FastW f, float *nn_in) {
    BO_SHARED FwShared sh;
    BO_SHARED int s_link[BO_FW_PATH_CAP];
    BO_SHARED unsigned s_flags[BO_FW_PATH_CAP];
    const int L = f.L, g = bo_block() / L, r = bo_block() % L, lane = bo_lane();
    if (e.phase[g] != PH_RUN || r >= f.n_rows[g]) return;
    const size_t ro = (size_t)g * L + r;
    const WRec *A = fw_arena(f, g);
    const DPos *BP = fw_bpos(f, g);
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
    const DPos P = make_move(BP[f.row_prun[ro]], fw_moves(f, g)[slot]);
    bool chk;
    const int n = bo_movegen(P, sh.moves, &chk);
    const int s = f.row_sim[ro];
    const int *path = f.sim_path + ((size_t)g * L + s) * BO_FW_PATH_CAP;
    const int d = f.sim_plen[(size_t)g * L + s] - 1;  // path[0..d], path[d] = this leaf
    const int t = terminal_eval_with(e, g, BP, P, sh.moves, n, chk, sh.moves2, sh.chain, [&]() {
        // ancestors path[k], k < d, are expanded: their positions head their runs.  Included while every move between
        // them and the leaf is reversible (python-chess pops back to the last irreversible move).
        for (int k = lane; k < d; k += 64) {
            const int lk = A[path[k]].link & BO_FW_LINK_MASK;
            s_link[k] = lk;
            s_flags[k] = BP[lk].flags;
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
                if (k - lo < BO_CHAIN_CAP) { sh.chain.hash[k - lo] = BP[s_link[k]].khash; sh.chain.ref[k - lo] = s_link[k]; }
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
        const int first = link & BO_FW_LINK_MASK, n = e.root_nlegal[g];
        const double *nz = e.noise + (size_t)g * BO_MAX_MOVES;
        for (int j = lane; j < n; j += 64) {
            WRec *c = A + (size_t)first * BO_FW_C + j;
            const float a = e.c.keep * c->prior;
            c->prior = (float)((double)a + e.c.eps * nz[j]);
        }
    }
    if (lane == 0) { e.phase[g] = PH_RUN; e.sims_done[g] = 0; f.n_rows[g] = 0; f.n_step[g] = 0; }
}

// fresh tree (one unexpanded root record) for the game slots that were (re)set up
BO_KERNEL void bo_k_fw_reset(Eng e, FastW f, const int *slots) {
    const int g = slots[bo_block()], lane = bo_lane();
    WRec *A = f.arena[0] + (size_t)g * f.NB * BO_FW_C;
    if (lane < BO_FW_C) {
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
    const WRec *S = f.arena[c0] + (size_t)g * f.NB * BO_FW_C;
    const bo_mv *SM = f.amove[c0] + (size_t)g * f.NB * BO_FW_C;
    const DPos *SP = f.bpos[c0] + (size_t)g * f.NB;
    WRec *D = f.arena[c0 ^ 1] + (size_t)g * f.NB * BO_FW_C;
    bo_mv *DM = f.amove[c0 ^ 1] + (size_t)g * f.NB * BO_FW_C;
    DPos *DP = f.bpos[c0 ^ 1] + (size_t)g * f.NB;
    // the played move among the old root's children
    int child = -1;
    const int rl = S[0].link;
    if (reuse && rl >= 0) {
        const int first = rl & BO_FW_LINK_MASK, nb = fw_nblk(rl);
        for (int i0 = 0; i0 < nb * BO_FW_C; i0 += 64) {
            const int i = i0 + lane;
            const bool hit = i < nb * BO_FW_C && S[(size_t)first * BO_FW_C + i].n >= 0 && SM[(size_t)first * BO_FW_C + i] == m;
            const uint64_t b = bo_ballot(hit);
            if (b) { child = first * BO_FW_C + i0 + bo_lsb64(b); break; }
        }
    }
    const int clink = child >= 0 ? S[child].link : FW_UNVISITED;
    const int cn = (child >= 0 && clink >= 0) ? S[child].n : 0;
    if (lane < BO_FW_C) {
        WRec c;
        c.n = lane == 0 ? cn : -1; c.w = 0.0f; c.prior = lane == 0 ? 1.0f : 0.0f; c.link = FW_UNVISITED;
        D[lane] = c;
    }
    int top = 1, flags = 0;
    bo_sync();
    if (clink >= 0) {
        // copy run `src_link` to D at block `top`; returns the new link
        #define BO_FW_COPY_RUN(src_link, new_link)                                                              \
        {                                                                                                       \
            const int _sf = (src_link) & BO_FW_LINK_MASK, _nb = fw_nblk(src_link);                             \
            for (int _i = lane; _i < _nb * BO_FW_C; _i += 64) {                                                 \
                D[(size_t)top * BO_FW_C + _i] = S[(size_t)_sf * BO_FW_C + _i];                                  \
                DM[(size_t)top * BO_FW_C + _i] = SM[(size_t)_sf * BO_FW_C + _i];                                \
            }                                                                                                   \
            if (lane == 0) DP[top] = SP[_sf];                                                                   \
            (new_link) = top | ((_nb - 1) << 24);                                                               \
            top += _nb;                                                                                         \
        }
        int nl;
        BO_FW_COPY_RUN(clink, nl)
        if (lane == 0) D[0].link = nl;
        bo_sync();
        for (int scan = 1; scan < top; scan++) {
            WRec c;
            c.n = -1; c.link = FW_UNVISITED;
            if (lane < BO_FW_C) c = D[(size_t)scan * BO_FW_C + lane];
            int link = c.link;
            uint64_t todo = bo_ballot(lane < BO_FW_C && c.n >= 0 && link >= 0);
            const bool mine = lane < BO_FW_C && c.n >= 0 && link >= 0;
            while (todo) {
                const int l = bo_lsb64(todo);
                todo &= todo - 1;
                const int sl = bo_readlane(link, l);
                int nl2 = FW_UNVISITED;
                if (top + fw_nblk(sl) <= f.NB) BO_FW_COPY_RUN(sl, nl2)
                else flags |= ST_NODE_OVERFLOW;  // cannot happen with the arena sizing of bo_engine_create; the subtree is dropped
                if (lane == l) link = nl2;
            }
            if (mine) D[(size_t)scan * BO_FW_C + lane].link = link;
            bo_sync();
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
    const int nch = link >= 0 ? n : 0, first = link >= 0 ? (link & BO_FW_LINK_MASK) : 0;
    const bo_mv *mv = e.root_moves + (size_t)g * BO_MAX_MOVES;
    int *ridx = e.res_idx + (size_t)g * BO_RES_CAP;
    float *rval = e.res_val + (size_t)g * BO_RES_CAP;
    int tot = 0, bv = -1, bk = 0x7fffffff;
    for (int i = lane; i < nch; i += 64) {
        const int v = A[(size_t)first * BO_FW_C + i].n;
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
            const int v = i < nch ? A[(size_t)first * BO_FW_C + i].n : 0;
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
