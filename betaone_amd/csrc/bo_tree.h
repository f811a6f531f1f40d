// betaone_amd/csrc/bo_tree.h -- the MCTS tree kernels (device code), one 64-lane wavefront per game.
//
// Replaces, for thousands of concurrent games at once, the reference's per-game Python objects:
//   MCTSNode.select_child / expand / update_recursive      /root/reference/mcts.py:45-144
//   run_mcts (root init, simulation loop, pi/best move)     /root/reference/mcts.py:155-280
//   _evaluate_batch (expand + backup of the pending rows)   /root/reference/mcts.py:283-295
//   utils.encode_board (leaf planes)                        /root/reference/utils.py:111-217
// with results bit-identical to the reference in dtype regime R3: every tree
// operation is a single binary32 operation, compiled with -ffp-contract=off and correctly rounded
// division; double->float roundings of math.sqrt()/int() come from host-built lookup tables.
//
// Data layout (HBM, structure of arrays, one contiguous block of NCAP nodes per game):
//   n_visits i32 | q f32 | prior f32           <- the 12 B/child the PUCT scan reads (coalesced,
//                                                  children of a node are contiguous)
//   parent i32 | first_child i32 | n_children i32 | move u16 | term i8 | eval_slot i16 | pos DPos(80 B)
// plus per game: the real game's position stack (draw rules, history planes), the repetition
// tracker, the pending-row run list and the cache of evaluated-but-not-yet-expanded leaves.
//
// Schedule (differs from the reference on purpose, results identical):
//   * the reference replays ONE leaf up to 96 times per NN batch (mcts.py:210-254); here a leaf is
//     evaluated once, the identical re-selections are recorded as a run (leaf, count), and the
//     flush replays the runs in order: children are created once with the final widened count and
//     every ancestor applies `count` incremental-mean updates in the reference's rounding order;
//   * one NN row per game per step (row index == game slot), so the net always sees a static
//     [G,120,8,8] batch and the step is capturable in a hipGraph.
#pragma once
#include "bo_chess.h"
#include <math.h>

#define BO_CH_CAP 32     // max children created per expansion (int(WIDEN*sqrt(BATCH)) must fit)
#define BO_PATH_CAP 1024 // max tree depth
#define BO_RES_CAP 256
#define BO_PLANES 120
#ifndef BO_PROF_SLOTS
#define BO_PROF_SLOTS 16  // bo_debug_profile: u64 counters per game (public: include/betaone_lab.h)
#endif
#define BO_ROW (BO_PLANES * 64)

enum { PH_IDLE = 0, PH_RUN = 1, PH_DONE = 2 };
enum { ST_OK = 0, ST_NODE_OVERFLOW = 1, ST_DEPTH_OVERFLOW = 2, ST_NAN_SCORE = 4, ST_PLY_OVERFLOW = 8,
       ST_ILLEGAL_ACTION = 16, ST_UL_OVERFLOW = 32, ST_TRK_OVERFLOW = 64 };
enum { POLICY_NONE = 0, POLICY_LOGITS = 1, POLICY_PROBS = 2 };

struct EngCfg {
    int G, S, B;                 // games, simulations per move, MCTS batch size
    int NCAP, PLY_CAP, TRK_CAP;  // nodes per game, positions per game, tracker entries per game
    int nstride;                 // distance between two games' node blocks in the node arrays (= NCAP; a view of ONE game's block has 0)
    int UL_MAX, CH_MAX;          // cached evaluated leaves per game; children per expansion
    float cpuct, keep;           // f32(CPUCT), f32(1 - DIRICHLET_EPSILON)
    double eps;                  // DIRICHLET_EPSILON
    int use_noise;               // DIRICHLET_ALPHA > 0
    int root_m;                  // children admitted by one root expand call: int(WIDEN*sqrt(1))
    int profile;                 // bo_debug_profile: accumulate per-phase shader cycles (s_memtime) per game
    int burst_two;               // terminal_burst keeps the previous burst's path in a second register set (0: off -- BETAONE_BURST_TWO_PATHS=0, for A/B tests)
};

struct Eng {
    EngCfg c;
    const float *sqrt_lut;  // [S+2]  f32(math.sqrt(n + 1e-8))
    const float *rcp_lut;   // [S+3]  RN(1 / n) (entry 0 unused): the divisor table of bo_div_count
    const int *widen_lut;   // [B+1]  int(WIDEN*sqrt(k)) for k rows (0 -> "all legal moves")
    // per game scalars
    int *phase, *sims_done, *n_nodes, *rows, *n_runs, *n_ul, *req_node, *req_nlegal, *status;
    int *ply, *trk_n, *n_hist, *ctx_mode, *root_nlegal, *root_term, *root_nch;
    int *stat_evals, *stat_flushes, *stat_term_sims, *stat_levels, *stat_children_scanned;
    // game stack, tracker, explicit history
    DPos *gpos, *trk, *hist;
    int *trk_cnt;
    // nodes [G][NCAP]
    int *n_visits, *parent, *first_child, *n_children;
    float *q, *prior;
    bo_mv *move;
    signed char *term;
    short *eval_slot;
    DPos *npos;
    // pending rows / evaluated leaves
    int *run_leaf, *run_cnt;          // [G][B]
    int *ul_node, *ul_nlegal;         // [G][UL_MAX]
    float *ul_value;
    bo_mv *ul_move;                   // [G][UL_MAX][CH_CAP]
    float *ul_prior;
    bo_mv *req_moves, *root_moves;    // [G][256]
    int *root_child_rank;             // [G][2*CH_CAP] legal-order index of each root child
    double *noise;                    // [G][256]
    bo_mv *played;                    // [G][PLY_CAP]
    // search results
    int *res_n, *res_idx, *res_best_idx, *res_best_mv, *res_total;
    float *res_val;
    int watch_n;                      // words at `watch` that are OR-ed into the copy (1, or 2: bo_nn_b1_word's [timeout code | saturation flag])
    int *watch, *res_watch;           // bo_engine_watch: a device status word of the evaluate stage (NULL: none); the result kernels copy it behind
                                      // the result block, so the ply's one host round trip brings it along
    int *played_now;                  // [G] or NULL: bo_k_play notes the move it played (0: refused) -- read by the fast mode's re-rooting
    unsigned long long *prof;         // [G][BO_PROF_SLOTS] cycles: apply, select, first-visit (movegen+draw rules), terminal backups, encode, flush, total; steps,
                                      // loop iterations, first visits.  profile = N > 1 counts only game-steps longer than N cycles
};

#define NOFF(e, g) ((size_t)(g) * (size_t)(e).c.nstride)

// exp for the policy softmax: the hardware v_exp_f32 (exp2) path on gfx950, libm in the emulator build.  The
// in-kernel softmax (policy_kind LOGITS) is held to the north-star tolerance (1e-4), not to bit-equality.
BO_DEV float bo_expf(float x) {
#if defined(BO_WAVE_EMU)
    return expf(x);
#else
    return __expf(x);
#endif
}
struct bo_f4 { float x, y, z, w; };
// exp of the PARITY mode's row softmax: the accurate expf on both sides of the seam -- bo_k_heads_rows (bo_heads.h) writes
// expf(x - max) / sum, and the step kernel's own softmax (policy_kind LOGITS) produces the SAME BITS from the same logits: same exp,
// same order of the sum (below), same division.  That identity is what lets a run whose softmax is in the step kernel be compared
// game for game with a run whose probabilities were recorded behind bo_k_heads_rows and replayed through the oracle.
BO_DEV float bo_exp_row(float x) { return expf(x); }
// max and sum(exp(x - max)) over one 4672-float policy row: 1168 float4 over 64 lanes.  All 19 loads of a lane are
// issued before the first use (one memory round trip for the 18.7 KB row; as two dependent-looking loops the row
// cost ~40 serial misses = 25 k cycles of a 87 k-cycle game-step) and both passes run from registers.
// ORDER of the sum = bo_k_heads_rows' (256 threads, thread t takes float4 t + 256 u, adds its exps one by one, a 32..1 butterfly per
// 64 threads, then (w0 + w1) + (w2 + w3)): lane l plays threads l, l + 64, l + 128, l + 192 -- its float4 number i belongs to thread
// l + 64 (i & 3).
#if defined(BO_WAVE_EMU)
#define BO_UNROLL
#else
#define BO_UNROLL _Pragma("unroll")
#endif
constexpr int BO_ROW_N4 = BO_NUM_ACTIONS / 4, BO_ROW_IT = (BO_ROW_N4 + 63) / 64;
BO_DEV float bo_wave_sum_desc_f(float v) {  // the head kernels' butterfly: partners 32, 16, .. 1
    for (int m = 32; m >= 1; m >>= 1) v = v + bo_shfl_xor_f(v, m);
    return v;
}
// v[] = this lane's float4s of the row (-inf beyond the end); returns max and sum(exp(x - max)) over the row
BO_DEV void row_load_max_sum(const float *row, bo_f4 (&v)[BO_ROW_IT], float *mx_out, float *sum_out) {
    const bo_f4 *r4 = reinterpret_cast<const bo_f4 *>(row);
    const int lane = bo_lane();
    const float ninf = -__builtin_inff();
    BO_UNROLL
    for (int i = 0; i < BO_ROW_IT; i++) {
        const int k = lane + 64 * i;
        v[i] = k < BO_ROW_N4 ? r4[k] : bo_f4{ninf, ninf, ninf, ninf};
    }
    float mx = ninf;
    BO_UNROLL
    for (int i = 0; i < BO_ROW_IT; i++) {
        const float a = v[i].x > v[i].y ? v[i].x : v[i].y, b = v[i].z > v[i].w ? v[i].z : v[i].w;
        const float c = a > b ? a : b;
        mx = c > mx ? c : mx;
    }
    mx = bo_wave_max_f(mx);
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    BO_UNROLL
    for (int i = 0; i < BO_ROW_IT; i++)
        if (lane + 64 * i < BO_ROW_N4) {
            float t = s[i & 3];
            t += bo_exp_row(v[i].x - mx); t += bo_exp_row(v[i].y - mx); t += bo_exp_row(v[i].z - mx); t += bo_exp_row(v[i].w - mx);
            s[i & 3] = t;
        }
    BO_UNROLL
    for (int w = 0; w < 4; w++) s[w] = bo_wave_sum_desc_f(s[w]);
    *mx_out = mx;
    *sum_out = (s[0] + s[1]) + (s[2] + s[3]);
}
BO_DEV void row_max_sum(const float *row, float *mx_out, float *sum_out) {
    bo_f4 v[BO_ROW_IT];
    row_load_max_sum(row, v, mx_out, sum_out);
}

BO_DEV int bo_uniform(int v) {
#if defined(BO_WAVE_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// ---- chain walk: the positions python-chess revisits when it pops back to the last irreversible
// move (Board.is_repetition / can_claim_threefold_repetition).  Calls f(const DPos&) per entry.
#define BO_CHAIN_CAP 192  // a reversible chain is shorter than the halfmove clock (< 150 where this is called)
struct ChainBuf {
    unsigned long long hash[BO_CHAIN_CAP];
    int ref[BO_CHAIN_CAP];  // node index (>= 0) or -1 - ply of the game history
};
// ref >= 0: a tree position, fetched by the caller's pos_of(ref); ref < 0: ply -1 - ref of the game's position stack
template <class PosFn> BO_DEV DPos chain_entry(const Eng &e, int g, PosFn pos_of, int ref) {
    return ref >= 0 ? pos_of(ref) : e.gpos[(size_t)g * e.c.PLY_CAP + (-1 - ref)];
}
// the game-history part of a chain (entries ply-1 .. J, J = last irreversible ply): found and read by all lanes at once
BO_DEV int chain_collect_history(const Eng &e, int g, ChainBuf &cb, int cnt) {
    const DPos *gp = e.gpos + (size_t)g * e.c.PLY_CAP;
    const int lane = bo_lane(), ply = e.ply[g];
    int J = 0;  // largest j in [1, ply] whose position was reached by an irreversible move, else 0
    for (int base = ply; base > 0; base -= 64) {
        const int j = base - lane;
        const bool irr = j > 0 && (gp[j].flags & F_IRREV);
        const unsigned long long m = bo_ballot(irr);
        if (m) { J = base - __builtin_ctzll(m); break; }
    }
    const int nh = ply - J;  // entries ply-1 .. J
    for (int i = lane; i < nh; i += 64) {
        if (cnt + i < BO_CHAIN_CAP) { const int j = ply - 1 - i; cb.hash[cnt + i] = gp[j].khash; cb.ref[cnt + i] = -1 - j; }
    }
    return cnt + nh;
}
// Collects (transposition hash, reference) of every chain entry into LDS and returns their number.  The in-tree part
// follows parent links (a few levels); the game-history part -- up to ~100 plies in drawn-out endgames, the tail
// of the step kernel when it was walked one dependent 80-byte load at a time -- is found and read by all lanes at
// once: ballot for the last irreversible ply, then one hash per lane.
// `path[0..d]` = the nodes root..leaf of the descent that has just ended in the leaf (select_leaf leaves it in LDS): ancestor
// path[k] belongs to the chain while no move between it and the leaf was irreversible.  All lanes read the path nodes'
// flags and hashes at once -- one memory round trip per 64 levels; following the parent links from the leaf took two
// dependent loads per level (a third of a first visit's time).
BO_DEV int chain_collect(const Eng &e, int g, const int *path, int d, ChainBuf &cb) {
    const size_t no = NOFF(e, g);
    const int lane = bo_lane();
    int cnt = 0;
    bool stop = false;
    for (int base = d; base >= 0 && !stop; base -= 64) {
        const int j = base - lane;  // this lane's path index, from the leaf (j = d) towards the root (j = 0)
        const bool in = j >= 0;
        const int node = in ? path[j] : 0;
        const uint32_t fl = in ? e.npos[no + node].flags : 0u;
        const uint32_t kh = in ? e.npos[no + node].khash : 0u;
        const uint64_t irr = bo_ballot(in && j >= 1 && (fl & F_IRREV));  // the root's own flag belongs to the game history
        const uint64_t below = BIT(lane) - 1;
        const bool take = in && j < d && (irr & below) == 0;  // an ancestor, and every node after it was reached reversibly
        const uint64_t tk = bo_ballot(take);
        if (take) {
            const int o = cnt + bo_popc64(tk & below);
            if (o < BO_CHAIN_CAP) { cb.hash[o] = kh; cb.ref[o] = node; }
        }
        cnt += bo_popc64(tk);
        stop = irr != 0;
    }
    if (!stop) cnt = chain_collect_history(e, g, cb, cnt);
    bo_sync();
    return cnt;
}

// Board.outcome(claim_draw=True) of a position P whose legal moves are mv[0..n): 0 ongoing, 1 mate, 2 draw.  `collect()`
// fills cb with the positions python-chess would revisit (the chain back to the last irreversible move) and returns their
// number; it is only called when the cheap rules have not decided; tree_pos(ref) = the position behind a ref >= 0 it stores.
template <class PosFn, class CollectFn>
BO_DEV int terminal_eval_with(const Eng &e, int g, PosFn tree_pos, const DPos &P, const bo_mv *mv, int n, bool in_check, bo_mv *scratch,
                              ChainBuf &cb, CollectFn collect) {
    if (n == 0 && in_check) return 1;
    if (insufficient_material(P)) return 2;
    if (n == 0) return 2;
    if (P.halfmove >= 150) return 2;
    const int npred = collect();
    const int nn = npred < BO_CHAIN_CAP ? npred : BO_CHAIN_CAP;
    int same = 0;
    for (int i = bo_lane(); i < nn; i += 64)
        if (cb.hash[i] == P.khash && key_equal(chain_entry(e, g, tree_pos, cb.ref[i]), P)) same++;
    const int self = 1 + bo_wave_sum(same);
    if (self >= 5) return 2;
    if (P.halfmove >= 100) return 2;
    if (P.halfmove >= 99) {  // can_claim_fifty_moves: some non-zeroing move reaches 100 without ending the game
        for (int j = 0; j < n; j++) {
            bo_mv m = mv[j];
            if (is_zeroing(P, m)) continue;
            DPos c = make_move(P, m);
            bool chk;
            int n2 = bo_movegen(c, scratch, &chk);
            if (n2 > 0) return 2;
        }
    }
    if (self >= 3) return 2;
    if (npred >= 3) {  // lookahead: a legal move into a position already seen twice
        bool hit = false;
        for (int j0 = 0; j0 < n; j0 += 64) {
            int j = j0 + bo_lane();
            bool act = j < n;
            bo_mv m = act ? mv[j] : (bo_mv)0;
            act = act && !is_irreversible(P, m);
            DPos c = P;
            if (act) c = make_move(P, m);
            int cc = 0;
            for (int i = 0; i < nn; i++)
                if (act && cb.hash[i] == c.khash && key_equal(chain_entry(e, g, tree_pos, cb.ref[i]), c)) cc++;
            if (bo_ballot(act && cc >= 2)) hit = true;
        }
        if (hit) return 2;
    }
    return 0;
}

// the same for the leaf path[d] of the reference-semantics tree
BO_DEV int terminal_eval(const Eng &e, int g, const int *path, int d, const DPos &P, const bo_mv *mv, int n, bool in_check, bo_mv *scratch,
                         ChainBuf &cb) {
    const DPos *np = e.npos + NOFF(e, g);
    return terminal_eval_with(e, g, [np](int ref) { return np[ref]; }, P, mv, n, in_check, scratch, cb, [&]() { return chain_collect(e, g, path, d, cb); });
}

// tracker.repetitions(board) = max(0, count - 1)   (utils.py:91-99)
BO_DEV int tracker_reps(const Eng &e, int g, const DPos &P) {
    const DPos *t = e.trk + (size_t)g * e.c.TRK_CAP;
    const int *tc = e.trk_cnt + (size_t)g * e.c.TRK_CAP;
    int n = e.trk_n[g], c = 0;
    for (int j = bo_lane(); j < n; j += 64)
        if (t[j].khash == P.khash && key_equal(t[j], P)) c += tc[j];
    c = bo_wave_sum(c);
    return c > 1 ? c - 1 : 0;
}

// one 14-plane history block (utils.py:172-188); lane == square
BO_DEV void encode_block(float *row, int block, const DPos &H, int rep) {
    const int s = bo_lane();
    float *p = row + (size_t)block * 14 * 64 + s;
#pragma unroll
    for (int t = 0; t < 6; t++) {
        p[(2 * t) * 64] = (float)((H.bb[t] & H.bb[BB_WHITE]) >> s & 1);
        p[(2 * t + 1) * 64] = (float)((H.bb[t] & H.bb[BB_BLACK]) >> s & 1);
    }
    p[12 * 64] = rep >= 1 ? 1.0f : 0.0f;
    p[13 * 64] = rep >= 2 ? 1.0f : 0.0f;
}
// planes 112..119 (utils.py:190-215)
BO_DEV void encode_scalars(float *row, const DPos &P) {
    const int s = bo_lane();
    float *p = row + 112 * 64 + s;
    p[0] = (P.flags & F_TURN) ? 1.0f : 0.0f;
    p[64] = (P.flags & 0x02u) ? 1.0f : 0.0f;
    p[128] = (P.flags & 0x04u) ? 1.0f : 0.0f;
    p[192] = (P.flags & 0x08u) ? 1.0f : 0.0f;
    p[256] = (P.flags & 0x10u) ? 1.0f : 0.0f;
    p[320] = (float)P.halfmove;
    p[384] = (float)P.fullmove;
    p[448] = pos_ep(P) == s ? 1.0f : 0.0f;
}
// planes 0..97 of a search: the <=7 boards before the root; constant for the whole search (mcts.py:242)
BO_DEV void encode_static(const Eng &e, int g, float *row) {
    const int s = bo_lane();
    const int nh = e.n_hist[g];
    const DPos *h = e.hist + (size_t)g * 7;
    for (int pl = 0; pl < (7 - nh) * 14; pl++) row[pl * 64 + s] = 0.0f;
    for (int i = 0; i < nh; i++) {
        DPos H = h[i];
        encode_block(row, 7 - nh + i, H, tracker_reps(e, g, H));
    }
}
BO_DEV void encode_leaf(const Eng &e, int g, float *row, const DPos &P) {
    encode_block(row, 7, P, tracker_reps(e, g, P));
    encode_scalars(row, P);
}

// MCTSNode.select_child repeated down to a leaf (mcts.py:218-230, 72-118)
// One PUCT descent (MCTSNode.select_child down to a leaf, mcts.py:72-118).  Per level ONE dependent memory round trip:
// the children's (visits, prior, q) and, with them, each child's own (first_child, n_children), so the chosen child's
// block is known without touching it; the parent-side visit count of the reference (mcts.py:89: the visits of the
// node ABOVE the one being expanded, the root's own at the top) is carried down in registers; sqrt comes from the
// table in the same round trip; the argmax over <= 16 children is four DPP exchanges (more children: two ds steps on top).
BO_DEV int select_leaf(const Eng &e, int g, int *flags, int *path, int *depth_out, const float *lut) {
    const size_t no = NOFF(e, g);
    const int lane = bo_lane();
    int cur = 0, levels = 0, scanned = 0;
    if (lane == 0) path[0] = 0;
    int nc = e.n_children[no], fc = e.first_child[no];
    int pv = e.n_visits[no], pv_next = pv;  // level 0 and level 1 both see the root's visits
    for (;;) {
        if (nc == 0) break;
        const float sp = lut[pv];
        float score = -__builtin_inff();
        int n = 0, c_fc = 0, c_nc = 0;
        if (lane < nc) {
            n = e.n_visits[no + fc + lane];
            c_fc = e.first_child[no + fc + lane];
            c_nc = e.n_children[no + fc + lane];
            const float p = e.prior[no + fc + lane];
            const float t1 = e.c.cpuct * p;
            const float t2 = t1 * sp;
            const float q_child = e.q[no + fc + lane];  // requested with the rest (a load behind `n > 0` is a second round trip)
            float qv = 0.0f, u = t2;
            if (n > 0) {
                qv = q_child;
                u = t2 / (float)(1 + n);
            }
            score = qv + u;
            if (!(score == score)) score = -__builtin_inff();  // NaN never wins `score > best`
        }
        int bi = lane;
#define BO_SEL_STEP(os_expr, oi_expr)                                                   \
        {                                                                               \
            const float os = (os_expr);                                                 \
            const int oi = (oi_expr);                                                   \
            if (os > score || (os == score && oi < bi)) { score = os; bi = oi; }        \
        }
        BO_SEL_STEP(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, score), 0)), BO_ROW_XCHG(bi, 0))
        BO_SEL_STEP(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, score), 1)), BO_ROW_XCHG(bi, 1))
        BO_SEL_STEP(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, score), 2)), BO_ROW_XCHG(bi, 2))
        BO_SEL_STEP(__builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, score), 3)), BO_ROW_XCHG(bi, 3))
        if (nc > 16) {  // rows -> wave (lanes >= nc hold -inf)
            BO_SEL_STEP(bo_shfl_xor_f(score, 16), bo_shfl_xor(bi, 16))
            BO_SEL_STEP(bo_shfl_xor_f(score, 32), bo_shfl_xor(bi, 32))
        }
#undef BO_SEL_STEP
        score = __builtin_bit_cast(float, bo_readlane(__builtin_bit_cast(int, score), 0));
        bi = bo_readlane(bi, 0);
        if (!(score > -__builtin_inff())) { *flags |= ST_NAN_SCORE; bi = 0; }  // reference: random.choice
        cur = fc + bi;
        levels++;
        if (lane == 0) path[levels] = cur;
        scanned += nc;
        pv = pv_next;
        pv_next = bo_readlane(n, bi);
        fc = bo_readlane(c_fc, bi);
        nc = bo_readlane(c_nc, bi);
        if (levels >= BO_PATH_CAP - 1) { *flags |= ST_DEPTH_OVERFLOW; break; }
    }
    if (lane == 0) { e.stat_levels[g] += levels; e.stat_children_scanned[g] += scanned; }
    *depth_out = levels;
    bo_sync();
    return cur;
}

// Repeated terminal simulations (mcts.py:235-238) without memory round trips.  A terminal leaf absorbs a simulation,
// its path is updated, and the next descent very often ends in the same leaf (a mating move, a claimable draw): in
// late games a search can spend hundreds of simulations this way, one dependent-load chain each.  Here the children
// of every node on the path (<= 4 levels x <= 16 children) are held in registers, 16 lanes per level: each
// simulation is applied to them in the reference's order of operations, the descent is re-evaluated from the
// registers, and as long as it reproduces the same path the loop continues.  Returns the number of simulations
// applied (>= 1); the caller re-selects from memory afterwards.  Bit-identical to backup_run + select_leaf.
#define BO_BURST_LEVELS 4
#define BO_BURST_FITS(S) (2 * (S) + 5 <= BO_NUM_ACTIONS)  // both tables in the step kernel's 18 KB probability buffer (S <= 2333)
// x / fn for fn = (float)n, n a small positive integer (a visit count), y = RN(1 / fn) from the host-built table: quotient
// estimate, exact remainder (fma), one correction (fma) -- the correctly rounded quotient (Markstein's division; it holds
// whenever the quotient is a normal number: checked against true division on 1.6e9 operand pairs, n <= 4100), in 3
// operations instead of the ~10 of the general sequence.  Tiny and non-finite quotients take the general division.
BO_DEV_NOINLINE float bo_div_general(float x, float fn) { return x / fn; }  // out of line: the compiler must branch around it
BO_DEV float bo_div_count(float x, float fn, float y) {
    const float q0 = x * y;
    const float r = __builtin_fmaf(-fn, q0, x);
    float q = __builtin_fmaf(r, y, q0);
    const float a = q0 < 0.0f ? -q0 : q0;
    if (x == 0.0f) q = q0;  // +-0 / n = +-0 (the common case: a child whose mean already equals the value backed up)
    else if (!(a >= 1e-30f && a <= 1e30f)) q = bo_div_general(x, fn);  // (never taken in a search: |quotient| in [1e-30, 1e30])
    return q;
}
// TWO paths: terminal simulations often ALTERNATE between two leaves under different children of an upper node (two mating
// lines; a mate and a claimable draw).  With one path in registers every such switch ended the burst after ~2 simulations
// and cost a descent from memory plus a burst set-up (48 of each per launch: the step kernel's p99).  `pb` / `db` / `vb`, when
// db > 0, is the path of the PREVIOUS burst of this launch: its children are held in a second register set below the level
// where the two paths part (the levels above are the same nodes: one set), and when the node at that level turns from one
// path's child to the other's -- and the other path's deeper levels still choose what they chose -- the sets are swapped and
// the loop goes on.  Everything a simulation does to the statistics is the same operation in the same order as before.
BO_DEV int terminal_burst(const Eng &e, int g, const int *path, int d, float v, int sims_left, float *lds, bool *staged,
                          const int *pb = nullptr, int db = 0, float vb = 0.0f, int *n_swaps = nullptr) {
    const size_t no = NOFF(e, g);
    const int lane = bo_lane(), grp = lane >> 4, j = lane & 15;
    // common nodes of the two paths: path[0 .. c-1] == pb[0 .. c-1]; they part at level c - 1 (the children of path[c - 1])
    int c = 0;
    if (db > 0) {
        while (c <= d && c <= db && path[c] == pb[c]) c++;
        if (!(c >= 1 && c <= d - 1 && c <= db - 1)) db = 0;  // identical, or parting at a last level (the sibling logic's case): one path
    }
    bool two = db > 0;
    int nv[BO_BURST_LEVELS + 1];  // uniform copies of the path nodes' visit counts
    int fc[BO_BURST_LEVELS], ncs[BO_BURST_LEVELS], chosen[BO_BURST_LEVELS];
#pragma unroll
    for (int k = 0; k <= BO_BURST_LEVELS; k++) nv[k] = k <= d ? e.n_visits[no + path[k]] : 0;
#pragma unroll
    for (int k = 0; k < BO_BURST_LEVELS; k++) {
        fc[k] = k < d ? e.first_child[no + path[k]] : 0;
        ncs[k] = k < d ? e.n_children[no + path[k]] : 0;
        chosen[k] = k < d ? path[k + 1] - fc[k] : -1;
    }
    int my_fc = 0, my_nc = 0, my_chosen = -1;  // this lane's child: group = level, j = child index
    // (the other path's loads are issued side by side with this path's: two dependent round trips for both, not four)
    int b_fc = 0, b_nc = 0, b_chosen = -1;
    if (two && grp < db) { b_fc = e.first_child[no + pb[grp]]; b_nc = e.n_children[no + pb[grp]]; }
    int b_pv = (two && grp >= 1 && grp <= db) ? e.n_visits[no + pb[grp - 1]] : 0;  // (used by levels > c only: up to level c the parent-side node is common)
#pragma unroll
    for (int k = 0; k < BO_BURST_LEVELS; k++)
        if (grp == k) { my_fc = fc[k]; my_nc = ncs[k]; my_chosen = chosen[k]; }
    if (two && grp < db) b_chosen = pb[grp + 1] - b_fc;
    if (two && bo_ballot(grp < db && b_nc > 16) != 0) { two = false; db = 0; }  // the other path no longer fits 16 lanes per level
    bool have = grp < d && j < my_nc;
    const bool last = grp == d - 1;  // the level whose children are leaves of the current path
    int cn = have ? e.n_visits[no + my_fc + j] : 0;
    float cq = have ? e.q[no + my_fc + j] : 0.0f;
    const float cp = have ? e.prior[no + my_fc + j] : 0.0f;
    // at the last level any child that is ALREADY KNOWN to be a terminal leaf may take the next simulation
    int cterm = (have && last && e.n_children[no + my_fc + j] == 0) ? (int)e.term[no + my_fc + j] : -1;
    // the other path's register set: levels >= c (its own nodes); its chosen child also at level c - 1 (same node, other child)
    int b_cn = 0, b_cterm = -1;
    float b_cq = 0.0f, b_cp = 0.0f;
    bool b_have = false;
    if (two) {
        b_have = grp >= c && grp < db && j < b_nc;
        b_cn = b_have ? e.n_visits[no + b_fc + j] : 0;
        b_cq = b_have ? e.q[no + b_fc + j] : 0.0f;
        b_cp = b_have ? e.prior[no + b_fc + j] : 0.0f;
        b_cterm = (b_have && grp == db - 1 && e.n_children[no + b_fc + j] == 0) ? (int)e.term[no + b_fc + j] : -1;
    }
    float v_cur = v, v_oth = vb;
    int d_oth = db;
    int done = 0;
    // Per simulated visit the loop needs sqrt(parent visits) of every level and the reciprocal of the visited child's new
    // count.  Both tables (S + 2 and S + 3 floats) are staged in LDS once per launch by all lanes (one memory round trip);
    // as global loads inside the loop their L2 latency (~700 cycles) bounded every visit.  (The caller takes the general
    // path for searches whose tables do not fit the 18 KB buffer: BO_BURST_FITS.)
    const int S = e.c.S;
    if (!*staged) {
        bo_sync();
        for (int t = lane; t < 2 * S + 5; t += 64) lds[t] = t < S + 2 ? e.sqrt_lut[t] : e.rcp_lut[t - (S + 2)];
        bo_sync();
        *staged = true;
    }
#define BO_SQ(i) lds[(i) <= S + 1 ? (i) : S + 1]
#define BO_RC(i) lds[S + 2 + ((i) <= S + 2 ? (i) : S + 2)]
    int pvl = nv[0];  // parent-side visit count of this lane's level (mcts.py:89): the root's for levels 0 and 1
#pragma unroll
    for (int k = 1; k < BO_BURST_LEVELS; k++)
        if (grp == k) pvl = nv[k - 1];
    float t1 = e.c.cpuct * cp, b_t1 = e.c.cpuct * b_cp;
    float y1 = BO_RC(1 + cn);  // RN(1 / (1 + cn)): the divisor of this child's exploration term
    float b_y1 = BO_RC(1 + b_cn);
    uint64_t last_bits = d > 0 ? (0xFFFFull << (16 * (d - 1))) : 0ull;
    bool last_l = last;
    int pv = pvl;  // parent-side visits of this lane's level on the CURRENT path; + 1 per simulation through it
    for (;;) {
        pv += 1;
        // (levels <= c: the parent-side node is common to both paths and `pv` counts for both; deeper levels keep one count per path)
        const float sp = BO_SQ(pv);
        // ---- apply one terminal simulation (MCTSNode.update along the path, mcts.py:120-144) ----
        if (have && j == my_chosen) {
            const float val = ((d - (grp + 1)) & 1) ? -v_cur : v_cur;  // the leaf sees v, its parent -v, ...
            cn += 1;
            const float dd = val - cq;
            const float ee = bo_div_count(dd, (float)cn, y1);  // == dd / (float)cn, correctly rounded
            cq = cq + ee;
            y1 = BO_RC(1 + cn);
        }
#pragma unroll
        for (int k = 0; k <= BO_BURST_LEVELS; k++) nv[k] += k <= d ? 1 : 0;
        // (the root's own q_value is not maintained here: nothing reads it -- select_child scores children only, the
        //  reference keeps it as a Python float that no code path looks at)
        done++;
        if (done >= sims_left) break;
        // ---- re-evaluate the descent from the registers (select_child at every level of the path) ----
        float score = -__builtin_inff();
        if (have) {
            const float t2 = t1 * sp;
            float qv = 0.0f, u = t2;
            if (cn > 0) { qv = cq; u = bo_div_count(t2, (float)(1 + cn), y1); }
            score = qv + u;
            if (!(score == score)) score = -__builtin_inff();
        }
        // Does every level still choose the child it chose?  One cross-lane read of the chosen child's score per lane and a
        // ballot (the full argmax costs four DPP rounds on two values; it is only needed when the LAST level moves on).
        const float cs = bo_shfl_f(score, (lane & 48) | (my_chosen & 15));
        const bool beat = have && (score > cs || (score == cs && j < my_chosen));
        const bool dead = grp < d && j == 0 && !(cs > -__builtin_inff());  // the chosen child's own score is NaN
        const uint64_t mb = bo_ballot(beat), md = bo_ballot(dead);
        if (md != 0) break;
        if ((mb & ~last_bits) != 0) {
            // an upper level turns elsewhere.  With the other path in registers: is it the level where the two part, does it turn
            // to the other path's child, and does the other path from there on still choose what it chose?
            if (!two) break;  // the next descent needs the general loop
            const uint64_t upper = mb & ~last_bits;
            const int kb = (int)(__builtin_ctzll(upper) >> 4);  // lowest level that turns
            if (kb != c - 1) break;
            int bi = j;
            float bs = score;
#define BO_BURST_STEP(kind)                                                                          \
            {                                                                                        \
                const float os = __builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, bs), kind)); \
                const int oi = BO_ROW_XCHG(bi, kind);                                                \
                if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }                          \
            }
            BO_BURST_STEP(0) BO_BURST_STEP(1) BO_BURST_STEP(2) BO_BURST_STEP(3)
#undef BO_BURST_STEP
            const int nb = bo_readlane(bi, 16 * (c - 1));            // the child level c - 1 now chooses
            const int ob = bo_readlane(b_chosen, 16 * (c - 1));      // the other path's child there
            if (nb != ob) break;
            // the other path below: every level must still choose its recorded child (its statistics are as it left them; the
            // parent-side count of level c is the common node's: `pv`)
            float bscore = -__builtin_inff();
            if (b_have) {
                const float spb = BO_SQ(grp == c ? pv : b_pv);
                const float t2 = b_t1 * spb;
                float qv = 0.0f, u = t2;
                if (b_cn > 0) { qv = b_cq; u = bo_div_count(t2, (float)(1 + b_cn), b_y1); }
                bscore = qv + u;
                if (!(bscore == bscore)) bscore = -__builtin_inff();
            }
            const float bcs = bo_shfl_f(bscore, (lane & 48) | (b_chosen & 15));
            const bool bbeat = b_have && (bscore > bcs || (bscore == bcs && j < b_chosen));
            const bool bdead = grp >= c && grp < d_oth && j == 0 && !(bcs > -__builtin_inff());
            if (bo_ballot(bbeat) != 0 || bo_ballot(bdead) != 0) break;
            // ---- swap the register sets: the other path is the current one from here on ----
            if (grp >= c) {
                { const int t_ = cn; cn = b_cn; b_cn = t_; }
                { const float t_ = cq; cq = b_cq; b_cq = t_; }
                { const float t_ = t1; t1 = b_t1; b_t1 = t_; }
                { const float t_ = y1; y1 = b_y1; b_y1 = t_; }
                { const bool t_ = have; have = b_have; b_have = t_; }
                { const int t_ = cterm; cterm = b_cterm; b_cterm = t_; }
                { const int t_ = my_fc; my_fc = b_fc; b_fc = t_; }
                { const int t_ = my_nc; my_nc = b_nc; b_nc = t_; }
            }
            if (grp > c) { const int t_ = pv; pv = b_pv; b_pv = t_; }
            if (grp >= c - 1) { const int t_ = my_chosen; my_chosen = b_chosen; b_chosen = t_; }
            { const int t_ = d; d = d_oth; d_oth = t_; }
            { const float t_ = v_cur; v_cur = v_oth; v_oth = t_; }
            last_l = grp == d - 1;
            last_bits = 0xFFFFull << (16 * (d - 1));
            if (n_swaps) *n_swaps += 1;
            continue;
        }
        if ((mb & last_bits) != 0) {  // the last level moves to a sibling: fine if that one is a known terminal leaf too
            int bi = j;
            float bs = score;
#define BO_BURST_STEP(kind)                                                                          \
            {                                                                                        \
                const float os = __builtin_bit_cast(float, BO_ROW_XCHG(__builtin_bit_cast(int, bs), kind)); \
                const int oi = BO_ROW_XCHG(bi, kind);                                                \
                if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }                          \
            }
            BO_BURST_STEP(0) BO_BURST_STEP(1) BO_BURST_STEP(2) BO_BURST_STEP(3)
#undef BO_BURST_STEP
            int bterm = (j == (bi & 15)) ? cterm : -2;  // term of the child this group selected: row maximum of the one candidate
            { int o = BO_ROW_XCHG(bterm, 0); bterm = o > bterm ? o : bterm; }
            { int o = BO_ROW_XCHG(bterm, 1); bterm = o > bterm ? o : bterm; }
            { int o = BO_ROW_XCHG(bterm, 2); bterm = o > bterm ? o : bterm; }
            { int o = BO_ROW_XCHG(bterm, 3); bterm = o > bterm ? o : bterm; }
            const int src = 16 * (d - 1);  // lane 0 of the last level's group holds its argmax
            const int nb = bo_readlane(bi, src), nt = bo_readlane(bterm, src);
            if (!(nt > 0)) break;
            if (last_l) my_chosen = nb;
            v_cur = nt == 1 ? 1.0f : 0.0f;
        }
    }
#undef BO_SQ
#undef BO_RC
    // ---- write the path(s) back ----
    if (have) { e.n_visits[no + my_fc + j] = cn; e.q[no + my_fc + j] = cq; }
    if (two && b_have) { e.n_visits[no + b_fc + j] = b_cn; e.q[no + b_fc + j] = b_cq; }
    if (lane == 0) e.n_visits[no] = nv[0];
    bo_sync();
    return done;
}

// MCTSNode.update_recursive applied `cnt` times with the same leaf value (mcts.py:120-144).  known_depth >= 0: known[0 ..
// known_depth] is the descent root..leaf that select_leaf has just left in LDS (no walk up the parent links: one dependent
// load per level saved); else the path is collected into `scratch`.
BO_DEV void backup_run(const Eng &e, int g, int leaf, float v, int cnt, int *scratch, int *flags, const int *known = nullptr,
                       int known_depth = -1) {
    const size_t no = NOFF(e, g);
    int d = 0;
    if (known_depth >= 0) {
        d = known_depth + 1;
    } else {
        for (int x = leaf; x >= 0 && d < BO_PATH_CAP; x = e.parent[no + x]) {
            if (bo_lane() == 0) scratch[d] = x;
            d++;
        }
        bo_sync();
    }
    for (int i = bo_lane(); i < d; i += 64) {  // i = distance from the leaf
        const int nd = known_depth >= 0 ? known[known_depth - i] : scratch[i];
        const float val = (i & 1) ? -v : v;
        int n = e.n_visits[no + nd];
        float qv = e.q[no + nd];
        // (The compiler's general division stays: both cheaper-looking forms of the quotient measured SLOWER in this loop -- RN(1 / n) by a second
        // division beside the chain + bo_div_count: flush 18.7 k -> 25.8 k cycles per game-step; RN(1 / n) from the host-built table, requested
        // eight iterations ahead, + bo_div_count: 22.0 k.  bo_div_count's guards (zero dividend, tiny quotient -> general division) cost
        // more per iteration than the general sequence they replace.  profiles/r05_device_turn_and_tiles.md section 7.)
        for (int c = 0; c < cnt; c++) {
            n += 1;
            const float dd = val - qv;
            const float ee = dd / (float)n;
            qv = qv + ee;
        }
        e.n_visits[no + nd] = n;
        e.q[no + nd] = qv;
    }
    bo_sync();
}

BO_DEV void init_node(const Eng &e, size_t no, int idx, int par, float prior, bo_mv m, const DPos &P) {
    e.n_visits[no + idx] = 0;
    e.q[no + idx] = 0.0f;
    e.prior[no + idx] = prior;
    e.parent[no + idx] = par;
    e.first_child[no + idx] = 0;
    e.n_children[no + idx] = 0;
    e.move[no + idx] = m;
    e.term[no + idx] = -1;
    e.eval_slot[no + idx] = -1;
    e.npos[no + idx] = P;
}

// _evaluate_batch's per-row expand + backup for all pending rows (mcts.py:291-295)
BO_DEV void flush_pending(const Eng &e, int g, int n_runs, int n_ul, int *n_nodes_io, int *path, int *flags, int cur_leaf = -1,
                          const int *cur_path = nullptr, int cur_depth = -1) {
    const size_t no = NOFF(e, g);
    const int lane = bo_lane();
    const int *rl = e.run_leaf + (size_t)g * e.c.B, *rc = e.run_cnt + (size_t)g * e.c.B;
    int n_nodes = *n_nodes_io;
    for (int u = 0; u < n_ul; u++) {
        const size_t uo = (size_t)g * e.c.UL_MAX + u;
        const int leaf = e.ul_node[uo];
        int k = 0;
        for (int r = 0; r < n_runs; r++) k += rl[r] == leaf ? rc[r] : 0;
        int c = e.widen_lut[k];             // int(WIDEN_COEFF*sqrt(k)) of the leaf's last row (mcts.py:55-57)
        const int nl = e.ul_nlegal[uo];
        if (c <= 0 || c > nl) c = nl;
        if (c > e.c.CH_MAX) c = e.c.CH_MAX;
        if (n_nodes + c > e.c.NCAP) { *flags |= ST_NODE_OVERFLOW; c = e.c.NCAP - n_nodes; }
        if (lane < c) {
            const DPos P = e.npos[no + leaf];
            const bo_mv m = e.ul_move[uo * BO_CH_CAP + lane];
            init_node(e, no, n_nodes + lane, leaf, e.ul_prior[uo * BO_CH_CAP + lane], m, make_move(P, m));
        }
        if (lane == 0) { e.first_child[no + leaf] = n_nodes; e.n_children[no + leaf] = c; }
        n_nodes += c;
    }
    bo_sync();
    for (int r = 0; r < n_runs; r++) {
        const int leaf = rl[r];
        const int slot = e.eval_slot[no + leaf];
        const bool known = leaf == cur_leaf && cur_depth >= 0;  // (the usual case: the batch's only leaf is the one just selected)
        backup_run(e, g, leaf, e.ul_value[(size_t)g * e.c.UL_MAX + slot], rc[r], path, flags, known ? cur_path : nullptr, known ? cur_depth : -1);
    }
    *n_nodes_io = n_nodes;
    if (lane == 0) e.stat_flushes[g] += 1;
}

// numpy's pairwise float32 sum of exactly 4672 contiguous values (np.ndarray.sum, mcts.py:201).
// numpy halves n (rounded down to a multiple of 8) until a block has <= 128 elements:
// 4672 -> 2336 -> 1168 -> 584 -> (288, 296) -> (144,144 | 144,152) -> (72,72 | 72,72 | 72,72 | 72,80),
// i.e. 64 leaf blocks at offsets 584*c + 72*k (the 8th of every group has 80 elements), each summed
// with 8 strided accumulators, combined by a balanced binary tree -- one leaf per lane + a butterfly.
BO_DEV float np_sum_4672(const float *a) {
    const int L = bo_lane();
    const float *b = a + (L >> 3) * 584 + (L & 7) * 72;
    const int n = (L & 7) == 7 ? 80 : 72;
    float r0 = b[0], r1 = b[1], r2 = b[2], r3 = b[3], r4 = b[4], r5 = b[5], r6 = b[6], r7 = b[7];
    for (int i = 8; i < n; i += 8) {
        r0 += b[i]; r1 += b[i + 1]; r2 += b[i + 2]; r3 += b[i + 3];
        r4 += b[i + 4]; r5 += b[i + 5]; r6 += b[i + 6]; r7 += b[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (int m = 1; m < 64; m <<= 1) res = res + bo_shfl_xor_f(res, m);
    return res;
}

// stable descending rank of p[j] among p[0..n): #greater + #equal-before (sorted(..., reverse=True), mcts.py:58-62)
BO_DEV int stable_rank(const float *p, int n, int j) {
    const float pj = p[j];
    int r = 0;
    for (int i = 0; i < n; i++) r += (p[i] > pj) || (p[i] == pj && i < j);
    return r;
}

struct StepShared {
    bo_mv moves[BO_MAX_MOVES];
    bo_mv moves2[BO_MAX_MOVES];
    float pv[BO_MAX_MOVES];
    int path[BO_PATH_CAP];
    int path2[BO_PATH_CAP];  // scratch of parent-link walks (path keeps the last descent)
    int bpath[8];            // the path of the launch's previous terminal burst (terminal_burst's second path)
    float probs[BO_NUM_ACTIONS];
    int rank_of[2 * BO_CH_CAP];
    ChainBuf chain;
};

// softmax row -> LDS (policy_kind LOGITS) or copy (PROBS)
BO_DEV void load_probs(const float *row, int kind, float *out) {
    const int lane = bo_lane();
    if (kind == POLICY_PROBS) {
        for (int i = lane; i < BO_NUM_ACTIONS; i += 64) out[i] = row[i];
    } else {  // the row stays in registers between the reduction and the normalised write
        float mx, sum;
        bo_f4 v[BO_ROW_IT];
        row_load_max_sum(row, v, &mx, &sum);
        BO_UNROLL
        for (int i = 0; i < BO_ROW_IT; i++) {
            const int k = lane + 64 * i;
            if (k < BO_ROW_N4) {
                out[4 * k] = bo_exp_row(v[i].x - mx) / sum; out[4 * k + 1] = bo_exp_row(v[i].y - mx) / sum;
                out[4 * k + 2] = bo_exp_row(v[i].z - mx) / sum; out[4 * k + 3] = bo_exp_row(v[i].w - mx) / sum;
            }
        }
    }
    bo_sync();
}

// root initialisation after the root's evaluation (mcts.py:185-203)
BO_DEV void apply_root(const Eng &e, int g, const float *row, int kind, StepShared &sh, int *n_nodes_io, int *flags) {
    const size_t no = NOFF(e, g);
    const int lane = bo_lane();
    const int n = e.root_nlegal[g];
    const bo_mv *mv = e.root_moves + (size_t)g * BO_MAX_MOVES;
    load_probs(row, kind, sh.probs);
    int m = e.c.root_m < n ? e.c.root_m : n;
    if (m > BO_CH_CAP) m = BO_CH_CAP;
    for (int j = lane; j < n; j += 64) sh.pv[j] = sh.probs[move_to_index(mv[j])];
    if (lane < 2 * BO_CH_CAP) sh.rank_of[lane] = -1;
    bo_sync();
    for (int j = lane; j < n; j += 64) {  // first expand: top-m raw priors
        int r = stable_rank(sh.pv, n, j);
        if (r < m) sh.rank_of[r] = j;
    }
    bo_sync();
    const DPos P = e.npos[no];
    int nch = m;
    if (lane < m) {
        int j = sh.rank_of[lane];
        init_node(e, no, 1 + lane, 0, sh.pv[j], mv[j], make_move(P, mv[j]));
        e.root_child_rank[(size_t)g * 2 * BO_CH_CAP + lane] = j;
    }
    bo_sync();
    if (e.c.use_noise) {  // mcts.py:190-201
        const double *nz = e.noise + (size_t)g * BO_MAX_MOVES;
        for (int j = lane; j < n; j += 64) {
            const int idx = move_to_index(mv[j]);
            const float a = e.c.keep * sh.probs[idx];
            const double s2 = (double)a + e.c.eps * nz[j];
            sh.probs[idx] = (float)s2;
        }
        bo_sync();
        const float sum = np_sum_4672(sh.probs);
        const float denom = sum + 1e-12f;
        for (int j = lane; j < n; j += 64) sh.pv[j] = sh.probs[move_to_index(mv[j])] / denom;
        bo_sync();
    }
    // second expand (mcts.py:203): top-m of the (noised) priors, children not yet present are appended
    for (int j = lane; j < n; j += 64) {
        int r = stable_rank(sh.pv, n, j);
        if (r < m) sh.rank_of[BO_CH_CAP + r] = j;
    }
    bo_sync();
    for (int r = 0; r < m; r++) {  // wave-uniform, m is tiny (1 with the default WIDEN_COEFF)
        const int j = sh.rank_of[BO_CH_CAP + r];
        bool present = false;
        for (int k = 0; k < m; k++) present = present || sh.rank_of[k] == j;
        if (!present) {
            if (lane == 0) {
                init_node(e, no, 1 + nch, 0, sh.pv[j], mv[j], make_move(P, mv[j]));
                e.root_child_rank[(size_t)g * 2 * BO_CH_CAP + nch] = j;
            }
            nch++;
        }
    }
    if (lane == 0) { e.first_child[no] = 1; e.n_children[no] = nch; e.root_nch[g] = nch; }
    *n_nodes_io = 1 + nch;
    bo_sync();
}

// The evaluate stage's TAIL inside the step (bo_step_heads): bo_k_heads_tiles has left the logits and the 16 K-chunk partial sums of
// value_fc1; what bo_k_heads_rows would do with a board's row -- softmax, value = tanh(value_fc2(relu(sum of the chunks + bias))) --
// the game's own wave does when (and only when) it consumes the row.  Same operations in the same order as bo_k_heads_rows
// (thread t of its 256 <-> lane t & 63, pass t >> 6): the value's bits are the ones that kernel would have written.
struct StepTail { const float *vpart, *b1, *w2, *b2; int rows; };  // vpart == NULL: the step reads value[g]
struct TailRegs { float part[4][16], b1[4], w2[4], b2; };
BO_DEV void tail_load(const StepTail &t, int g, TailRegs &r) {  // requests only: issued in front of the row's loads, one round trip for both
    const int lane = bo_lane();
    BO_UNROLL
    for (int w = 0; w < 4; w++) {
        BO_UNROLL
        for (int ks = 0; ks < 16; ks++) r.part[w][ks] = t.vpart[((size_t)ks * t.rows + g) * 256 + lane + 64 * w];
        r.b1[w] = t.b1[lane + 64 * w];
        r.w2[w] = t.w2[lane + 64 * w];
    }
    r.b2 = t.b2[0];
}
BO_DEV float tail_value(const TailRegs &r) {
    float s[4];
    BO_UNROLL
    for (int w = 0; w < 4; w++) {
        float h = 0.0f;
        BO_UNROLL
        for (int ks = 0; ks < 16; ks++) h += r.part[w][ks];
        h += r.b1[w];
        s[w] = (h > 0.0f ? h : 0.0f) * r.w2[w];
    }
    for (int m = 32; m >= 1; m >>= 1) {  // the four butterflies side by side
        BO_UNROLL
        for (int w = 0; w < 4; w++) s[w] = s[w] + bo_shfl_xor_f(s[w], m);
    }
    return tanhf(((s[0] + s[1]) + (s[2] + s[3])) + r.b2);
}

// a leaf's evaluation arrives: keep its value and its CH_MAX best (move, prior) pairs until the flush
BO_DEV void apply_leaf(const Eng &e, int g, int leaf, const float *row, int kind, const float *value_of, const StepTail &vt, StepShared &sh,
                       int *n_ul_io, int *flags) {
    const size_t no = NOFF(e, g);
    const int lane = bo_lane();
    const int n = e.req_nlegal[g];
    const bo_mv *mv = e.req_moves + (size_t)g * BO_MAX_MOVES;
    int slot = *n_ul_io;
    if (slot >= e.c.UL_MAX) { *flags |= ST_UL_OVERFLOW; slot = e.c.UL_MAX - 1; }
    const size_t uo = (size_t)g * e.c.UL_MAX + slot;
    float value;
    if (kind == POLICY_PROBS) {
        value = value_of[g];
        for (int j = lane; j < n; j += 64) sh.pv[j] = row[move_to_index(mv[j])];
    } else {  // softmax restricted to what is needed: max and sum over the row, exp at the legal indices
        float mx, sum;
        if (vt.vpart) {  // bo_step_heads: the value's operands are requested in front of the row (one round trip for both)
            TailRegs tr;
            tail_load(vt, g, tr);
            row_max_sum(row, &mx, &sum);
            value = tail_value(tr);
        } else {
            value = value_of[g];
            row_max_sum(row, &mx, &sum);
        }
        for (int j = lane; j < n; j += 64) sh.pv[j] = bo_exp_row(row[move_to_index(mv[j])] - mx) / sum;
    }
    bo_sync();
    const int M = n < e.c.CH_MAX ? n : e.c.CH_MAX;
    for (int j = lane; j < n; j += 64) {
        int r = stable_rank(sh.pv, n, j);
        if (r < M) { e.ul_move[uo * BO_CH_CAP + r] = mv[j]; e.ul_prior[uo * BO_CH_CAP + r] = sh.pv[j]; }
    }
    if (lane == 0) {
        e.ul_node[uo] = leaf;
        e.ul_nlegal[uo] = n;
        e.ul_value[uo] = value;
        e.eval_slot[no + leaf] = (short)slot;
    }
    *n_ul_io = slot + 1;
    bo_sync();
}

// ---- kernels -------------------------------------------------------------------------------------

// One step of every running search: consume the previous NN outputs (row g <-> game g), advance the
// simulation loop until the game needs a new evaluation or its search is complete, and write the
// requested leaf's planes into NN input row g.
// (returns the game's node count when it leaves)
BO_DEV int step_body(const Eng &e, int g, const float *policy, const float *value, int kind, float *nn_in, StepShared &sh, const StepTail &vt) {
    const int lane = bo_lane();
    const size_t no = NOFF(e, g);
    int flags = 0;
    int sims = e.sims_done[g], rows = e.rows[g], n_runs = e.n_runs[g], n_ul = e.n_ul[g], n_nodes = e.n_nodes[g];
    int req = e.req_node[g];
    int *rl = e.run_leaf + (size_t)g * e.c.B, *rc = e.run_cnt + (size_t)g * e.c.B;
    float *row = nn_in + (size_t)g * BO_ROW;

    const bool prof = e.c.profile != 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_iter = 0, n_first = 0, n_burst = 0, n_burst_sims = 0, n_general = 0, n_swaps = 0;
    unsigned long long tk = prof ? bo_clock() : 0ull;
    const unsigned long long t_start = tk;
#define BO_PROF(slot)                                                    \
    if (prof) { const unsigned long long _n = bo_clock(); pc[slot] += _n - tk; tk = _n; }
    if (req >= 0 && kind == POLICY_NONE) return n_nodes;  // evaluation still outstanding
    const float *lut = e.sqrt_lut;  // (an LDS copy costs a round trip per launch; the table entry is requested with the children)
    if (req >= 0) {
        if (req == 0) apply_root(e, g, policy + (size_t)g * BO_NUM_ACTIONS, kind, sh, &n_nodes, &flags);
        else apply_leaf(e, g, req, policy + (size_t)g * BO_NUM_ACTIONS, kind, value, vt, sh, &n_ul, &flags);
        req = -1;
    }
    BO_PROF(0)
    int phase = PH_RUN;
    bool burst_tables_staged = false;  // sh.probs is free from here on: terminal_burst keeps its two tables there
    // Terminal simulations need no evaluation, so a game that runs into hundreds of them (a mating move, a claimable draw:
    // up to the whole search) would keep its wavefront busy for ~1000 cycles each while every other game of the launch has
    // long asked for its next evaluation -- the launch, and with it the whole ply, waited for that one game.  A launch
    // therefore absorbs at most MCTS_BATCH_SIZE of them per game and then YIELDS: the game asks for nothing (its NN row is
    // evaluated and ignored) and carries on in the next launch.  Results are unchanged (the same operations in the same
    // order), and the search needs no more launches than one that spends those simulations on a batch of 96 rows would.
    int term_budget = e.c.B;
    int b_depth = 0;      // depth of sh.bpath (0: none)
    float b_tv = 0.0f;    // its leaf's terminal value
    int path_leaf = -1, path_depth = -1;  // the leaf whose descent sh.path holds (none yet in this launch)
    for (;;) {
        if (sims >= e.c.S) {  // mcts.py:256-257
            if (rows > 0) { flush_pending(e, g, n_runs, n_ul, &n_nodes, sh.path2, &flags, path_leaf, sh.path, path_depth); rows = n_runs = n_ul = 0; }
            BO_PROF(5)
            phase = PH_DONE;
            break;
        }
        int depth;
        const int leaf = select_leaf(e, g, &flags, sh.path, &depth, lut);
        path_leaf = leaf; path_depth = depth;  // sh.path[0..depth] = root..leaf until something else uses the buffer
        n_iter++;
        BO_PROF(1)
        int t = e.term[no + leaf];
        if (t < 0) {  // first visit: legal moves + is_terminal()  (mcts.py:235, cached per node)
            n_first++;
            const DPos P = e.npos[no + leaf];
            bool chk;
            const int n = bo_movegen(P, sh.moves, &chk);
            t = terminal_eval(e, g, sh.path, depth, P, sh.moves, n, chk, sh.moves2, sh.chain);
            if (lane == 0) e.term[no + leaf] = (signed char)t;
            if (t == 0) {  // it will be evaluated now: keep its ordered legal moves for apply_leaf
                for (int j = lane; j < n; j += 64) e.req_moves[(size_t)g * BO_MAX_MOVES + j] = sh.moves[j];
                if (lane == 0) e.req_nlegal[g] = n;
            }
            bo_sync();
            BO_PROF(2)
        }
        if (t > 0) {  // mcts.py:235-238: terminal leaves absorb the simulation, no NN row
            const float tv = t == 1 ? 1.0f : 0.0f;
            int applied = 1;
            bool small = depth <= BO_BURST_LEVELS && BO_BURST_FITS(e.c.S);
            for (int k = 0; k < depth && small; k++) small = e.n_children[no + sh.path[k]] <= 16;
            BO_PROF(6)
            const int may = e.c.S - sims < term_budget ? e.c.S - sims : term_budget;
            if (small) {
                const bool small_b = b_depth > 0 && e.c.burst_two != 0;  // the previous burst's path rides along (terminal_burst checks that it still fits)
                applied = terminal_burst(e, g, sh.path, depth, tv, may, sh.probs, &burst_tables_staged, sh.bpath, small_b ? b_depth : 0, b_tv, &n_swaps);
                n_burst++; n_burst_sims += applied;
                bo_sync();
                if (lane <= depth) sh.bpath[lane] = sh.path[lane];
                b_depth = depth; b_tv = tv;
                bo_sync();
            }
            else { backup_run(e, g, leaf, tv, 1, sh.path2, &flags, sh.path, depth); n_general++; }  // deep or wide path: one simulation the general way
            BO_PROF(7)
            sims += applied;
            term_budget -= applied;
            if (lane == 0) e.stat_term_sims[g] += applied;
            BO_PROF(3)
            if (term_budget <= 0 && sims < e.c.S) break;  // yield: req stays -1, phase stays RUN
            continue;
        }
        if (e.eval_slot[no + leaf] < 0) {  // needs the net: emit planes 98..119 into row g and pause
            encode_leaf(e, g, row, e.npos[no + leaf]);
            req = leaf;
            if (lane == 0) e.stat_evals[g] += 1;
            BO_PROF(4)
            break;
        }
        // mcts.py:247-254: until the next flush nothing changes, so every remaining row of this batch
        // (or of the simulation budget) re-selects this same leaf
        int cnt = e.c.B - rows;
        if (cnt > e.c.S - sims) cnt = e.c.S - sims;
        if (lane == 0) { rl[n_runs] = leaf; rc[n_runs] = cnt; }
        n_runs++;
        rows += cnt;
        sims += cnt;
        bo_sync();
        if (rows >= e.c.B) { flush_pending(e, g, n_runs, n_ul, &n_nodes, sh.path2, &flags, path_leaf, sh.path, path_depth); rows = n_runs = n_ul = 0; }
        BO_PROF(5)
    }
#undef BO_PROF
    if (prof && lane == 0) {
        const unsigned long long total = bo_clock() - t_start;
        if (e.c.profile == 1 || total > (unsigned long long)e.c.profile) {
            unsigned long long *pp = e.prof + (size_t)g * BO_PROF_SLOTS;
            pp[10] += (unsigned long long)n_burst; pp[11] += (unsigned long long)n_burst_sims; pp[12] += (unsigned long long)n_general;
            pp[13] += pc[6]; pp[14] += pc[7]; pp[15] += (unsigned long long)n_swaps;
            for (int i = 0; i < 6; i++) pp[i] += pc[i];
            pp[6] += total;
            pp[7] += 1;
            pp[8] += (unsigned long long)n_iter;
            pp[9] += (unsigned long long)n_first;
        }
    }
    if (lane == 0) {
        e.sims_done[g] = sims; e.rows[g] = rows; e.n_runs[g] = n_runs; e.n_ul[g] = n_ul; e.n_nodes[g] = n_nodes;
        e.req_node[g] = req; e.phase[g] = phase;
        if (flags) e.status[g] |= flags;
    }
    return n_nodes;
}
BO_KERNEL void bo_k_step(Eng e, const float *policy, const float *value, int kind, float *nn_in, StepTail vt) {
    BO_SHARED StepShared sh;
    const int g = bo_block();
    if (e.phase[g] != PH_RUN) return;
    step_body(e, g, policy, value, kind, nn_in, sh, vt);
}

// Prepare the root of game g's next search from the top of its position stack: reset the tree,
// generate the root's ordered legal moves, evaluate is_game_over(claim_draw=True).
BO_DEV int root_prepare(const Eng &e, int g, StepShared &sh) {  // returns the root's terminal code (every lane)
    const size_t no = NOFF(e, g);
    const int lane = bo_lane();
    const int ply = e.ply[g];
    const DPos P = e.gpos[(size_t)g * e.c.PLY_CAP + ply];
    if (lane == 0) {
        init_node(e, no, 0, -1, 1.0f, 0, P);
        e.n_nodes[g] = 1; e.sims_done[g] = 0; e.rows[g] = 0; e.n_runs[g] = 0; e.n_ul[g] = 0;
        e.req_node[g] = -1; e.phase[g] = PH_IDLE; e.root_nch[g] = 0;
        if (e.ctx_mode[g] == 0) {  // self-play context: history = the <=7 real positions before the root
            int nh = ply < 7 ? ply : 7;
            e.n_hist[g] = nh;
            for (int i = 0; i < nh; i++) e.hist[(size_t)g * 7 + i] = e.gpos[(size_t)g * e.c.PLY_CAP + ply - nh + i];
        }
    }
    bo_sync();
    bool chk;
    const int n = bo_movegen(P, sh.moves, &chk);
    if (lane == 0) sh.path[0] = 0;
    bo_sync();
    const int t = terminal_eval(e, g, sh.path, 0, P, sh.moves, n, chk, sh.moves2, sh.chain);
    for (int j = lane; j < n; j += 64) {
        e.root_moves[(size_t)g * BO_MAX_MOVES + j] = sh.moves[j];
        e.req_moves[(size_t)g * BO_MAX_MOVES + j] = sh.moves[j];
    }
    if (lane == 0) { e.root_nlegal[g] = n; e.req_nlegal[g] = n; e.root_term[g] = t; e.term[no] = (signed char)t; }
    bo_sync();
    return t;
}

// (re)build game stacks: start position + moves; computes key fields; prepares the first root.
struct SetupArgs {
    const int *slots;        // [n] game slots to set up
    const DPos *start;       // [n] raw start positions (key fields filled here)
    const bo_mv *moves;      // [n][max_moves]
    const int *n_moves;      // [n]
    int max_moves;
    const DPos *hist;        // [n][7] explicit history boards or NULL
    const int *n_hist;       // [n] (-1 = self-play context)
    const DPos *trk;         // [n][max_trk] explicit tracker keys or NULL
    const int *trk_cnt;      // [n][max_trk]
    const int *n_trk;        // [n]
    int max_trk;
};
BO_DEV void finish_key(DPos &p) {
    if (!(p.flags & F_EPKEY_MASK) && pos_ep(p) >= 0 && has_legal_ep(p)) p.flags |= (uint32_t)(pos_ep(p) + 1) << F_EPKEY_SHIFT;
    p.khash = key_hash(p);
}
BO_KERNEL void bo_k_setup(Eng e, SetupArgs a) {
    BO_SHARED StepShared sh;
    const int i = bo_block(), lane = bo_lane();
    const int g = a.slots[i];
    DPos *gp = e.gpos + (size_t)g * e.c.PLY_CAP;
    DPos *tk = e.trk + (size_t)g * e.c.TRK_CAP;
    int *tc = e.trk_cnt + (size_t)g * e.c.TRK_CAP;
    const bool explicit_ctx = a.n_hist && a.n_hist[i] >= 0;
    if (lane == 0) {
        DPos p = a.start[i];
        p.flags |= F_IRREV;
        finish_key(p);
        gp[0] = p;
        int ply = 0, st = 0;
        for (int k = 0; k < a.n_moves[i]; k++) {
            if (ply + 1 >= e.c.PLY_CAP) { st |= ST_PLY_OVERFLOW; break; }
            bo_mv m = a.moves[(size_t)i * a.max_moves + k];
            e.played[(size_t)g * e.c.PLY_CAP + ply] = m;
            gp[ply + 1] = make_move(gp[ply], m);
            ply++;
        }
        e.ply[g] = ply;
        e.status[g] = st;
        e.ctx_mode[g] = explicit_ctx ? 1 : 0;
        if (explicit_ctx) {
            int nh = a.n_hist[i];
            e.n_hist[g] = nh;
            for (int k = 0; k < nh; k++) { DPos h = a.hist[(size_t)i * 7 + k]; finish_key(h); e.hist[(size_t)g * 7 + k] = h; }
            int nt = a.n_trk[i] < e.c.TRK_CAP ? a.n_trk[i] : e.c.TRK_CAP;
            for (int k = 0; k < nt; k++) { DPos t = a.trk[(size_t)i * a.max_trk + k]; finish_key(t); tk[k] = t; tc[k] = a.trk_cnt[(size_t)i * a.max_trk + k]; }
            e.trk_n[g] = nt;
        } else {  // utils.RepetitionTracker.add_board after every real move (self_play.py:93,182)
            for (int k = 0; k <= ply; k++) { tk[k] = gp[k]; tc[k] = 1; }
            e.trk_n[g] = ply + 1;
        }
        e.stat_evals[g] = e.stat_flushes[g] = e.stat_term_sims[g] = e.stat_levels[g] = e.stat_children_scanned[g] = 0;
    }
    bo_sync();
    root_prepare(e, g, sh);
}

// Start the searches of all games flagged in `go`: static history planes into NN row g, phase = RUN.
BO_KERNEL void bo_k_search_begin(Eng e, const int *go, float *nn_in) {
    const int g = bo_block();
    if (!go[g]) return;
    encode_static(e, g, nn_in + (size_t)g * BO_ROW);
    if (bo_lane() == 0) e.phase[g] = PH_RUN;
}

// The same with the decision on the device: a wanted game whose root is not terminal searches (mcts.py:160-162).  The host
// learns the roots' state from a copy it does not wait for before the first evaluation is enqueued (bo_selfplay_turn, flag 4).
BO_KERNEL void bo_k_search_begin_want(Eng e, const int *want, float *nn_in) {
    const int g = bo_block();
    if (!want[g] || e.root_term[g] != 0) return;
    encode_static(e, g, nn_in + (size_t)g * BO_ROW);
    if (bo_lane() == 0) e.phase[g] = PH_RUN;
}

// End running searches early, between two steps: flush the pending rows as the tail batch (mcts.py:256-257), drop the
// outstanding evaluation request, phase = DONE.  The tree then is the reference's tree after sims_done simulations.
BO_KERNEL void bo_k_stop(Eng e, const int *mask) {
    BO_SHARED int path[BO_PATH_CAP];
    const int g = bo_block(), lane = bo_lane();
    if ((mask && !mask[g]) || e.phase[g] != PH_RUN) return;
    int flags = 0, n_nodes = e.n_nodes[g];
    if (e.rows[g] > 0) flush_pending(e, g, e.n_runs[g], e.n_ul[g], &n_nodes, path, &flags);
    bo_sync();
    if (lane == 0) {
        e.rows[g] = 0; e.n_runs[g] = 0; e.n_ul[g] = 0; e.n_nodes[g] = n_nodes;
        e.req_node[g] = -1; e.phase[g] = PH_DONE;
        if (flags) e.status[g] |= flags;
    }
}

// Two small blocks of words between device memory and PINNED, device-mapped host memory, by the compute queue itself: the turn of a
// ply (result block out, sampled actions and go flags in, root info out) then has no copy commands between its kernels -- each
// hipMemcpyAsync was ~10-50 us of runtime work plus a blit kernel of its own on the path the device idles through.  Either pair may be
// empty (n = 0).  Host memory written here is visible to the host once the stream / the event behind this kernel has been waited for.
BO_KERNEL void bo_k_ship(int *dst_a, const int *src_a, int n_a, int *dst_b, const int *src_b, int n_b, int n_blocks) {
    const int i0 = bo_block() * 64 + bo_lane(), stride = n_blocks * 64;
    for (int i = i0; i < n_a; i += stride) dst_a[i] = src_a[i];
    for (int i = i0; i < n_b; i += stride) dst_b[i] = src_b[i];
}

// pi and best move of a finished search (mcts.py:259-280)
BO_DEV void result_body(const Eng &e, int g) {
    const int lane = bo_lane();
    if (g == 0 && lane == 0) e.res_watch[0] = e.watch ? (e.watch[0] | (e.watch_n > 1 && e.watch[1] ? 0x10000 : 0)) : 0;
    if (e.phase[g] != PH_DONE) return;
    const size_t no = NOFF(e, g);
    const int nch = e.n_children[no], fc = e.first_child[no], n = e.root_nlegal[g];
    const bo_mv *mv = e.root_moves + (size_t)g * BO_MAX_MOVES;
    int *ridx = e.res_idx + (size_t)g * BO_RES_CAP;
    float *rval = e.res_val + (size_t)g * BO_RES_CAP;
    const int v = lane < nch ? e.n_visits[no + fc + lane] : 0;
    const int total = bo_wave_sum(v);
    if (total > 0) {
        // best = first maximum in LEGAL-move order (mcts.py:279), children carry their legal-order index
        int key = lane < nch ? e.root_child_rank[(size_t)g * 2 * BO_CH_CAP + lane] : 0x7fffffff;
        int bv = v, bk = key;
        for (int m = 1; m < 64; m <<= 1) {
            int ov = bo_shfl_xor(bv, m), ok = bo_shfl_xor(bk, m);
            if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        // a never-created child has 0 visits: if the maximum is 0 < total that cannot happen; bk is valid
        const uint64_t nzmask = bo_ballot(lane < nch && v > 0);
        if (lane < nch && v > 0) {
            int o = bo_popc64(nzmask & (BIT(lane) - 1));
            ridx[o] = move_to_index(e.move[no + fc + lane]);
            rval[o] = (float)((double)v / (double)total);
        }
        if (lane == 0) {
            e.res_n[g] = bo_popc64(nzmask);
            e.res_best_mv[g] = mv[bk];
            e.res_best_idx[g] = move_to_index(mv[bk]);
            e.res_total[g] = total;
        }
    } else {
        for (int j = lane; j < n; j += 64) { ridx[j] = move_to_index(mv[j]); rval[j] = (float)(1.0 / (double)n); }
        if (lane == 0) {
            e.res_n[g] = n;
            e.res_best_mv[g] = n ? mv[0] : 0;
            e.res_best_idx[g] = n ? move_to_index(mv[0]) : -1;  // -1: the reference raises ValueError
            e.res_total[g] = 0;
        }
    }
}
BO_KERNEL void bo_k_result(Eng e) { result_body(e, bo_block()); }

// Play the sampled action in every game with action[g] >= 0 (self_play.py:125-184), then prepare
// the next root.  action -2 = "play res_best_mv".
BO_DEV int play_body(const Eng &e, int g, int a, StepShared &sh) {  // returns the new root's terminal code, -1 if the root did not change
    const int lane = bo_lane();
    if (a == -1) return -1;
    const int ply = e.ply[g], tn = e.trk_n[g];  // read before lane 0 updates them further down
    DPos *gp = e.gpos + (size_t)g * e.c.PLY_CAP;
    const DPos P = gp[ply];
    const int n = e.root_nlegal[g];
    const bo_mv *mv = e.root_moves + (size_t)g * BO_MAX_MOVES;
    const bo_mv best = (bo_mv)e.res_best_mv[g];
    bo_mv m = best;
    if (a >= 0 && !index_to_move(a, P, &m)) m = best;  // self_play.py:127-137
    bool ok = false, best_ok = false;
    for (int j0 = 0; j0 < n; j0 += 64) {
        int j = j0 + lane;
        ok = ok || bo_ballot(j < n && mv[j] == m) != 0;
        best_ok = best_ok || bo_ballot(j < n && mv[j] == best) != 0;
    }
    if (e.played_now && lane == 0) e.played_now[g] = 0;
    if (!ok) {  // self_play.py:142-167
        if (best != m && best_ok) m = best;
        else { if (lane == 0) { e.status[g] |= ST_ILLEGAL_ACTION; e.phase[g] = PH_IDLE; } return -1; }
    }
    if (ply + 1 >= e.c.PLY_CAP || tn >= e.c.TRK_CAP) { if (lane == 0) e.status[g] |= ST_PLY_OVERFLOW; return -1; }
    if (lane == 0) {
        if (e.played_now) e.played_now[g] = m;
        const DPos c = make_move(P, m);
        gp[ply + 1] = c;
        e.played[(size_t)g * e.c.PLY_CAP + ply] = m;
        e.ply[g] = ply + 1;
        e.trk[(size_t)g * e.c.TRK_CAP + tn] = c;  // tracker.add_board (self_play.py:182)
        e.trk_cnt[(size_t)g * e.c.TRK_CAP + tn] = 1;
        e.trk_n[g] = tn + 1;
    }
    bo_sync();
    return root_prepare(e, g, sh);
}
BO_KERNEL void bo_k_play(Eng e, const int *action) {
    BO_SHARED StepShared sh;
    play_body(e, bo_block(), action[bo_block()], sh);
}

// ---- the ply's turn on the device (bo_selfplay_autoturn; self_play.py:121-184 + the next iteration's mcts.py:160-162) -------------------
// Result -> temperature sample -> play -> begin the next search, enqueued BEHIND the searches' last step: the device goes from the last
// tree step of a ply straight into the next ply's root evaluation, no host round trip in between.  What the host still owns is every
// random draw and every libm call: the uniform np.random.choice would draw for the move (self_play.py:73) is drawn ahead and handed in,
// and apply_temperature's p ** (1 / T) (self_play.py:37, NumPy -> libm pow) comes from a table the host built with the same libm for
// every visit count c = 0..S (pi = f32(c / S), mcts.py:273).  Everything else of select_move_with_temperature / RandomState.choice on a pi
// of <= 2 non-zero entries (E2: the reference's root has <= 2 children) is IEEE binary32 / binary64 arithmetic, restated operation
// by operation from csrc/bo_hostrng.h: hr_select_action.
struct TurnArgs {
    const double *u;      // [G] host memory (pinned, device-mapped): this game's choice() uniform
    const int *flags;     // [G] host memory: bit 0 the game searched this ply (sample + play), bit 1 its temperature is not 1 (fullmove >= threshold,
                          //     self_play.py:66): p ** (1 / T) from `pw`, bit 2 a next search is wanted
    const double *pw;     // [S + 1] device: pow((double)f32(c / S), 1 / T_final)
    int *action;          // [G] out: the sampled action index, -1 = the game did not search
    int *state;           // [G] out: 0 ok, 1 the game's search is still running, 2 a pi this sampler does not cover (the host raises), 3 the evaluate
                          //     stage's watched fault word is set (bo_engine_watch: no move is played from an invalid evaluation; the host raises)
    int *cres;            // [G][8] out: n, best action index, best move, total visits, index 0, index 1, value 0 bits, value 1 bits (pi order of bo_k_result)
};

// hr_select_action on (index, visit count) pairs; *ok = false where hr_select_action returns -1 (or the table does not apply)
BO_DEV int turn_sample(int n, int i0, int c0, int i1, int c1, int total, int S, bool power, const double *pw, double u, bool *ok) {
    *ok = true;
    if (n < 1 || n > 2 || total < 1) { *ok = false; return -1; }
    float p0 = (float)((double)c0 / (double)total), p1 = n == 2 ? (float)((double)c1 / (double)total) : 0.0f;  // mcts.py:273
    if (n == 2 && i1 < i0) { int t = i0; i0 = i1; i1 = t; float q = p0; p0 = p1; p1 = q; t = c0; c0 = c1; c1 = t; }
    if (power) {  // apply_temperature, self_play.py:37-45
        if (total != S) { *ok = false; return -1; }
        double s0 = pw[c0], s1 = n == 2 ? pw[c1] : 0.0;  // (non-finite powers were zeroed when the table was built)
        const double sum = s0 + s1;
        if (!(sum > 1e-9)) { *ok = false; return -1; }
        p0 = (float)(s0 / sum);
        p1 = (float)(s1 / sum);
        const float rs = p0 + p1;
        if (__builtin_fabsf(rs - 1.0f) > (float)1e-6 && rs > (float)1e-9) { p0 = p0 / rs; p1 = p1 / rs; }
    }
    const float prob_sum = p0 + p1;  // self_play.py:68-72
    if (__builtin_fabsf(prob_sum - 1.0f) > (float)1e-6) {
        if (prob_sum > (float)1e-9) { p0 = p0 / prob_sum; p1 = p1 / prob_sum; }
        else { *ok = false; return -1; }
    }
    const double d0 = (double)p0, d1 = (double)p1;  // RandomState.choice: cdf = cumsum(double(p)); cdf /= cdf[-1]; searchsorted(u, 'right')
    if (__builtin_fabs((d0 + d1) - 1.0) > 3.4e-4 || d0 < 0 || d1 < 0) { *ok = false; return -1; }
    const double k0 = d0, k1 = d0 + d1;
    const double last = n == 2 ? k1 : k0;
    if (n == 1) return i0;
    return (k0 / last > u) ? i0 : i1;
}

// first kernel of the turn: results of the finished searches + the sampled moves.  No wave changes any game's phase here, so
// `state` is a consistent picture of "did every search finish?" for the second kernel.
BO_KERNEL void bo_k_turn_sample(Eng e, TurnArgs a) {
    const int g = bo_block(), lane = bo_lane();
    result_body(e, g);
    const int fl = a.flags[g], ph = e.phase[g];
    int st = 0, act = -1, n = 0, ia = -1, ib = -1, ca = 0, cb = 0, total = 0;
    if (e.watch && (e.watch[0] != 0 || (e.watch_n > 1 && e.watch[1] != 0))) st = 3;
    else if (ph == PH_RUN) st = 1;
    else if ((fl & 1) && ph == PH_DONE) {
        const size_t no = NOFF(e, g);
        const int nch = e.n_children[no], fc = e.first_child[no];
        const int v = lane < nch ? e.n_visits[no + fc + lane] : 0;
        const int idx = lane < nch ? move_to_index(e.move[no + fc + lane]) : -1;
        total = bo_wave_sum(v);
        const uint64_t nz = bo_ballot(v > 0);
        n = bo_popc64(nz);
        if (n >= 1) { const int l0 = bo_lsb64(nz); ia = bo_shfl(idx, l0); ca = bo_shfl(v, l0); }
        if (n >= 2) { const int l1 = bo_lsb64(nz & (nz - 1)); ib = bo_shfl(idx, l1); cb = bo_shfl(v, l1); }
        bool ok;
        act = turn_sample(n, ia, ca, ib, cb, total, e.c.S, (fl & 2) != 0, a.pw, a.u[g], &ok);
        if (!ok) { st = 2; act = -1; }
    }
    if (lane == 0) {
        a.action[g] = act; a.state[g] = st;
        int *c = a.cres + (size_t)g * 8;
        c[0] = n; c[1] = e.res_best_idx[g]; c[2] = e.res_best_mv[g]; c[3] = total; c[4] = ia; c[5] = ib;
        c[6] = n >= 1 ? __builtin_bit_cast(int, (float)((double)ca / (double)total)) : 0;
        c[7] = n >= 2 ? __builtin_bit_cast(int, (float)((double)cb / (double)total)) : 0;
    }
}

// second kernel: nothing happens unless EVERY search had finished and every pi could be sampled (the host then steps once more / raises:
// bo_selfplay_autoturn_collect); else bo_k_play + bo_k_search_begin_want in one launch
BO_KERNEL void bo_k_turn_play(Eng e, TurnArgs a, float *nn_in) {
    BO_SHARED StepShared sh;
    const int g = bo_block(), lane = bo_lane();
    bool bad = false;
    for (int j0 = 0; j0 < e.c.G; j0 += 64) bad = bad || bo_ballot(j0 + lane < e.c.G && a.state[j0 + lane] != 0) != 0;
    if (bad) return;
    int t = play_body(e, g, a.action[g], sh);
    if (t < 0) t = e.root_term[g];  // (the root did not change in this launch)
    if (!(a.flags[g] & 4) || t != 0) return;
    encode_static(e, g, nn_in + (size_t)g * BO_ROW);
    if (lane == 0) e.phase[g] = PH_RUN;
}

// Final training encodings of a finished game (self_play.py:200-208): record i uses
// board_history[max(0,i-7) : i+1] and the END-OF-GAME tracker.  One wave per (game, ply).
BO_KERNEL void bo_k_encode_game(Eng e, int g, int first, float *out) {
    const int i = first + bo_block(), s = bo_lane();
    const DPos *gp = e.gpos + (size_t)g * e.c.PLY_CAP;
    float *row = out + (size_t)bo_block() * BO_ROW;
    const int h0 = i - 7 > 0 ? i - 7 : 0, nb = i + 1 - h0;
    for (int pl = 0; pl < (8 - nb) * 14; pl++) row[pl * 64 + s] = 0.0f;
    for (int k = 0; k < nb; k++) {
        DPos H = gp[h0 + k];
        encode_block(row, 8 - nb + k, H, tracker_reps(e, g, H));
    }
    encode_scalars(row, gp[i]);
}

// Expansion of a compact record (the wire format of betaone_amd/records.py) back into the dense training
// encodings of self_play.py:200-208, on ANY rank: pos[0..n_pos) are the game's positions, the tracker is
// the multiset of their keys (end-of-game tracker).  One wave per encoded ply.
BO_KERNEL void bo_k_encode_positions(const DPos *pos, int n_pos, int first, float *out) {
    const int i = first + bo_block(), s = bo_lane();
    float *row = out + (size_t)bo_block() * BO_ROW;
    const int h0 = i - 7 > 0 ? i - 7 : 0, nb = i + 1 - h0;
    for (int pl = 0; pl < (8 - nb) * 14; pl++) row[pl * 64 + s] = 0.0f;
    for (int k = 0; k < nb; k++) {
        const DPos H = pos[h0 + k];
        int c = 0;
        for (int j = s; j < n_pos; j += 64) c += key_equal(pos[j], H) ? 1 : 0;
        c = bo_wave_sum(c);
        encode_block(row, 8 - nb + k, H, c > 1 ? c - 1 : 0);
    }
    encode_scalars(row, pos[i]);
}

// Stand-alone batch utilities (tests, drop-in utils.encode_board): legal moves + outcome of positions
// given with their move stacks already loaded into game slots is covered by bo_k_setup; this one
// runs the move generator alone on raw positions.
BO_KERNEL void bo_k_movegen(const DPos *pos, bo_mv *out, int *n_out, int *check_out) {
    BO_SHARED bo_mv mv[BO_MAX_MOVES];
    const int i = bo_block(), lane = bo_lane();
    DPos P = pos[i];
    bool chk;
    const int n = bo_movegen(P, mv, &chk);
    for (int j = lane; j < n; j += 64) out[(size_t)i * BO_MAX_MOVES + j] = mv[j];
    if (lane == 0) { n_out[i] = n; check_out[i] = chk ? 1 : 0; }
}
