// betaone_amd/csrc/bo_fast.h -- FAST search mode.  NOT the reference's semantics.
//
// The reference's search has no virtual loss, so every NN batch holds one position repeated up to 96 times and the
// root never gets more than two children (mcts.py:186,203,210-254).  This mode keeps the
// reference's interfaces (same game stack, legal-move order, draw rules, input planes, (state, pi, z) records) but
// runs a conventional AlphaZero-style batched search, clearly labelled as diverging from the reference:
//   * L distinct leaves per game per step, selected one after another with a VIRTUAL LOSS (n += 1, W -= 1 on the
//     path) so that successive descents spread out; NN batch = G x L rows, row = g*L + r;
//   * full-width expansion: every legal move becomes a child, prior = policy mass renormalised over the legal
//     moves; Dirichlet noise on all root priors;
//   * backup with the sign alternating per ply and the virtual loss removed in the same pass (parent-chain writes
//     from one wavefront: lane k updates path node k, so the chain is written in parallel);
//   * W[node] is the value sum seen by the player who moved INTO the node, so PUCT needs no negation:
//     score = W/n + cpuct * P * sqrt(N_parent) / (1 + n).
// Arithmetic is plain binary32 in a fixed order; tests/fast_reference.py restates it in NumPy and the trees are
// compared bit for bit (there is no reference implementation to compare with).
#pragma once
#include "bo_tree.h"

#define BO_FAST_PATH_CAP 192

struct FastEng {
    int L;            // leaves per game per step
    int *n_rows;      // [G]   NN rows requested by the last step
    int *n_step_sims; // [G]   simulations selected by the last step (incl. terminal hits and collisions)
    int *row_leaf;    // [G][L] leaf node of row r
    int *row_nlegal;  // [G][L]
    bo_mv *row_moves; // [G][L][256] legal moves of row r's leaf (python-chess order)
    int *sim_row;     // [G][L] row whose value sim s backs up (-1: terminal, already backed up)
    int *sim_plen;    // [G][L]
    int *sim_path;    // [G][L][PATH_CAP] node ids root..leaf
};

// lane-strided partial sums + butterfly: the summation order tests/fast_reference.py mirrors
BO_DEV float fast_sum(const float *v, int n) {
    float a = 0.0f;
    for (int j = bo_lane(); j < n; j += 64) a = a + v[j];
    return bo_wave_sum_f(a);
}

// expand `leaf` with all its legal moves; priors pv[0..n) already normalised
BO_DEV bool fast_expand(const Eng &e, int g, int leaf, const bo_mv *mv, const float *pv, int n, int *n_nodes_io, int *flags) {
    const size_t no = NOFF(e, g);
    int first = *n_nodes_io;
    if (first + n > e.c.NCAP) { *flags |= ST_NODE_OVERFLOW; return false; }
    for (int j = bo_lane(); j < n; j += 64) {
        const size_t c = no + first + j;
        e.n_visits[c] = 0; e.q[c] = 0.0f; e.prior[c] = pv[j]; e.parent[c] = leaf; e.first_child[c] = 0; e.n_children[c] = 0;
        e.move[c] = mv[j]; e.term[c] = -1; e.eval_slot[c] = -2;  // -2: position not materialised yet
    }
    if (bo_lane() == 0) { e.first_child[no + leaf] = first; e.n_children[no + leaf] = n; }
    *n_nodes_io = first + n;
    return true;
}

// remove the virtual loss of one simulation and add its value: lane k handles path node k
BO_DEV void fast_backup(const Eng &e, int g, const int *path, int plen, float v_leaf_mover) {
    const size_t no = NOFF(e, g);
    for (int k = bo_lane(); k < plen; k += 64) {
        if (k == 0) continue;  // root: only its visit count matters (incremented at selection)
        const int nd = path[k];
        const float s = ((plen - 1 - k) & 1) ? -v_leaf_mover : v_leaf_mover;
        e.q[no + nd] = (e.q[no + nd] + 1.0f) + s;
    }
}

BO_KERNEL void bo_k_fast_step(Eng e, FastEng f, const float *policy, const float *value, int kind, float *nn_in) {
    BO_SHARED StepShared sh;
    const int g = bo_block(), lane = bo_lane();
    if (e.phase[g] != PH_RUN) return;
    const size_t no = NOFF(e, g);
    const int L = f.L;
    int flags = 0, n_nodes = e.n_nodes[g], sims = e.sims_done[g];
    int n_rows = f.n_rows[g], n_step = f.n_step_sims[g];
    int *row_leaf = f.row_leaf + (size_t)g * L, *row_nl = f.row_nlegal + (size_t)g * L;
    int *sim_row = f.sim_row + (size_t)g * L, *sim_plen = f.sim_plen + (size_t)g * L;
    int *sim_path = f.sim_path + (size_t)g * L * BO_FAST_PATH_CAP;
    bo_mv *row_moves = f.row_moves + (size_t)g * L * BO_MAX_MOVES;

    // ---- 1. consume the net's output for the rows of the previous step: expand + backup -------------------
    if (n_rows > 0) {
        if (kind == POLICY_NONE) return;
        for (int r = 0; r < n_rows; r++) {
            const int leaf = row_leaf[r], n = row_nl[r];
            const bo_mv *mv = row_moves + (size_t)r * BO_MAX_MOVES;
            const float *prow = policy + ((size_t)g * L + r) * BO_NUM_ACTIONS;
            if (kind == POLICY_PROBS) {
                for (int j = lane; j < n; j += 64) sh.pv[j] = prow[move_to_index(mv[j])];
            } else {  // softmax over the legal moves only
                float mx = -__builtin_inff();
                for (int j = lane; j < n; j += 64) { float x = prow[move_to_index(mv[j])]; sh.pv[j] = x; mx = x > mx ? x : mx; }
                mx = bo_wave_max_f(mx);
                for (int j = lane; j < n; j += 64) sh.pv[j] = bo_expf(sh.pv[j] - mx);
            }
            bo_sync();
            const float sum = fast_sum(sh.pv, n);
            for (int j = lane; j < n; j += 64) sh.pv[j] = sum > 0.0f ? sh.pv[j] / sum : 1.0f / (float)n;
            bo_sync();
            if (leaf == 0 && e.c.use_noise) {  // Dirichlet noise on every root prior
                const double *nz = e.noise + (size_t)g * BO_MAX_MOVES;
                for (int j = lane; j < n; j += 64) {
                    const float a = e.c.keep * sh.pv[j];
                    sh.pv[j] = (float)((double)a + e.c.eps * nz[j]);
                }
                bo_sync();
            }
            fast_expand(e, g, leaf, mv, sh.pv, n, &n_nodes, &flags);
            if (lane == 0) e.eval_slot[no + leaf] = -1;
            bo_sync();
        }
        if (row_leaf[0] == 0 && n_step == 0) {  // the root's own evaluation counts as its first visit
            if (lane == 0) e.n_visits[no] = 1;
        }
        for (int s = 0; s < n_step; s++) {
            const int r = sim_row[s];
            if (r < 0) continue;
            // value[] is from the leaf's side to move; the player who moved into the leaf sees -v
            fast_backup(e, g, sim_path + (size_t)s * BO_FAST_PATH_CAP, sim_plen[s], -value[(size_t)g * L + r]);
            bo_sync();
        }
        sims += n_step;
        n_rows = 0;
        n_step = 0;
        bo_sync();
    }

    // ---- 2. select up to L leaves with virtual loss ---------------------------------------------------------
    int phase = PH_RUN;
    if (e.n_children[no] == 0 && e.term[no] == 0) {  // root not expanded yet: its evaluation is row 0
        for (int j = lane; j < e.root_nlegal[g]; j += 64) row_moves[j] = e.root_moves[(size_t)g * BO_MAX_MOVES + j];
        encode_leaf(e, g, nn_in + (size_t)g * L * BO_ROW, e.npos[no]);
        if (lane == 0) { row_leaf[0] = 0; row_nl[0] = e.root_nlegal[g]; e.stat_evals[g] += 1; }
        n_rows = 1;
    } else if (e.term[no] != 0 || sims >= e.c.S) {
        phase = PH_DONE;
    } else {
        while (n_step < L && sims + n_step < e.c.S) {
            int *path = sim_path + (size_t)n_step * BO_FAST_PATH_CAP;
            int cur = 0, d = 1;
            if (lane == 0) { path[0] = 0; e.n_visits[no] += 1; }
            bo_sync();
            for (;;) {
                const int nc = e.n_children[no + cur];
                if (nc == 0 || d >= BO_FAST_PATH_CAP) break;
                const int fc = e.first_child[no + cur];
                const float sq = sqrtf((float)e.n_visits[no + cur]);
                float best = -__builtin_inff();
                int bi = 0x7fffffff;
                for (int i0 = 0; i0 < nc; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < nc) {
                        const int n = e.n_visits[no + fc + i];
                        const float w = e.q[no + fc + i], p = e.prior[no + fc + i];
                        const float t1 = e.c.cpuct * p;
                        const float t2 = t1 * sq;
                        const float u = t2 / (float)(1 + n);
                        const float qv = n > 0 ? w / (float)n : 0.0f;
                        const float sc = qv + u;
                        if (sc > best) { best = sc; bi = i; }
                    }
                }
                for (int m = 1; m < 64; m <<= 1) {
                    const float os = bo_shfl_xor_f(best, m);
                    const int oi = bo_shfl_xor(bi, m);
                    if (os > best || (os == best && oi < bi)) { best = os; bi = oi; }
                }
                if (bi >= nc) bi = 0;
                cur = bo_uniform(fc + bi);
                if (lane == 0) {
                    path[d] = cur;
                    e.n_visits[no + cur] += 1;      // virtual loss
                    e.q[no + cur] = e.q[no + cur] - 1.0f;
                }
                d++;
                bo_sync();
            }
            const int leaf = cur;
            int slot = e.eval_slot[no + leaf];
            int t = e.term[no + leaf];
            if (slot == -2) {  // first visit: materialise the position, legal moves, is_game_over(claim_draw=True)
                const DPos P = make_move(e.npos[no + e.parent[no + leaf]], e.move[no + leaf]);
                if (lane == 0) e.npos[no + leaf] = P;
                bo_sync();
                bool chk;
                const int n = bo_movegen(P, sh.moves, &chk);
                t = terminal_eval(e, g, leaf, P, sh.moves, n, chk, sh.moves2, sh.chain);
                slot = -1;
                if (lane == 0) { e.term[no + leaf] = (signed char)t; e.eval_slot[no + leaf] = -1; }
                if (t == 0 && n_rows < L) {  // becomes NN row n_rows
                    bo_mv *mv = row_moves + (size_t)n_rows * BO_MAX_MOVES;
                    for (int j = lane; j < n; j += 64) mv[j] = sh.moves[j];
                    encode_leaf(e, g, nn_in + ((size_t)g * L + n_rows) * BO_ROW, P);
                    if (lane == 0) { row_leaf[n_rows] = leaf; row_nl[n_rows] = n; e.eval_slot[no + leaf] = (short)n_rows; e.stat_evals[g] += 1; }
                    slot = n_rows;
                    n_rows++;
                }
                bo_sync();
            } else if (t == 0 && slot == -1 && e.n_children[no + leaf] == 0 && n_rows < L) {
                // a materialised, unexpanded, non-pending leaf (expansion was refused earlier): evaluate it again
                bool chk;
                const DPos P = e.npos[no + leaf];
                const int n = bo_movegen(P, sh.moves, &chk);
                bo_mv *mv = row_moves + (size_t)n_rows * BO_MAX_MOVES;
                for (int j = lane; j < n; j += 64) mv[j] = sh.moves[j];
                encode_leaf(e, g, nn_in + ((size_t)g * L + n_rows) * BO_ROW, P);
                if (lane == 0) { row_leaf[n_rows] = leaf; row_nl[n_rows] = n; e.eval_slot[no + leaf] = (short)n_rows; e.stat_evals[g] += 1; }
                slot = n_rows;
                n_rows++;
                bo_sync();
            }
            if (t > 0) {  // terminal: exact value now; mate = +1 for the player who delivered it
                fast_backup(e, g, path, d, t == 1 ? 1.0f : 0.0f);
                if (lane == 0) { sim_row[n_step] = -1; sim_plen[n_step] = d; e.stat_term_sims[g] += 1; }
                bo_sync();
            } else {
                // slot >= 0: (possibly shared) NN row; slot == -1 here means "no row left": value 0 for this visit
                if (slot < 0) fast_backup(e, g, path, d, 0.0f);
                if (lane == 0) { sim_row[n_step] = slot; sim_plen[n_step] = d; }
                bo_sync();
            }
            n_step++;
        }
        if (n_rows == 0) {  // only terminal hits this step: account for them now
            sims += n_step;
            n_step = 0;
            if (sims >= e.c.S) phase = PH_DONE;
        }
    }
    if (lane == 0) {
        e.sims_done[g] = sims; e.n_nodes[g] = n_nodes; e.phase[g] = phase;
        f.n_rows[g] = n_rows; f.n_step_sims[g] = n_step;
        e.req_node[g] = n_rows > 0 ? row_leaf[0] : -1;
        if (flags) e.status[g] |= flags;
    }
}

// planes 0..97 into all L rows of game g, phase = RUN
BO_KERNEL void bo_k_fast_search_begin(Eng e, FastEng f, const int *go, float *nn_in) {
    const int g = bo_block();
    if (!go[g]) return;
    for (int r = 0; r < f.L; r++) encode_static(e, g, nn_in + ((size_t)g * f.L + r) * BO_ROW);
    if (bo_lane() == 0) { e.phase[g] = PH_RUN; f.n_rows[g] = 0; f.n_step_sims[g] = 0; }
}

// pi over ALL legal root moves = child visits / total; best = first maximum in legal-move order
BO_KERNEL void bo_k_fast_result(Eng e) {
    const int g = bo_block(), lane = bo_lane();
    if (e.phase[g] != PH_DONE) return;
    const size_t no = NOFF(e, g);
    const int nch = e.n_children[no], fc = e.first_child[no], n = e.root_nlegal[g];
    const bo_mv *mv = e.root_moves + (size_t)g * BO_MAX_MOVES;
    int *ridx = e.res_idx + (size_t)g * BO_RES_CAP;
    float *rval = e.res_val + (size_t)g * BO_RES_CAP;
    int tot = 0, bv = -1, bk = 0x7fffffff;
    for (int i = lane; i < nch; i += 64) {
        const int v = e.n_visits[no + fc + i];
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
            const int v = i < nch ? e.n_visits[no + fc + i] : 0;
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
