// betaone_amd/csrc/bo_engine.cpp -- host side of the C ABI (include/betaone_engine.h).
// Compiled by hipcc for gfx950 into libbetaone_hip.so together with the kernels in bo_tree.h.
// No torch types, no chess logic on the host: FEN / UCI text is parsed into plain bitboards and
// everything else (replaying the move stack, keys, legality, draw rules) happens on the device.
#include "../../include/betaone_engine.h"
#include "../../include/betaone_lab.h"  // measurement / introspection entry points: exported by the same library, not part of the boundary

#include "bo_tree.h"
#include "bo_fastw.h"
#include "bo_select_wide.h"
#include "bo_replay.h"
#include "bo_nn_fused.h"
#include "bo_conv.h"
#include "bo_tower.h"
#include "bo_tower_wg.h"
#include "bo_tower_h.h"
#include "bo_tower_s.h"
#include "bo_tower_s16.h"
#include "bo_tower_h16.h"
#include "bo_heads.h"
#include "bo_tower_b1.h"
#include "bo_rt.h"
#include "bo_hostrng.h"

#include <math.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <algorithm>
#include <deque>

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define RT(call)                                                                                     \
    do {                                                                                             \
        int _rc = (call);                                                                            \
        if (_rc != 0) return fail(BO_E_HIP, std::string(#call) + ": " + rt_errstr(_rc));             \
    } while (0)

struct bo_engine {
    Eng d;
    FastW f;
    int fast_reuse = 1;  // keep the played child's subtree for the next search (bo_fast_options)
    // optional timing of the fast mode's select + backup kernel with HIP events on the launch stream (bo_fast_stats)
    int sel_profile = 0, sel_pending = 0;
    long long sel_launches = 0;
    double sel_ms = 0.0;
#if !defined(BO_WAVE_EMU)
    hipEvent_t sel_ev0 = nullptr, sel_ev1 = nullptr;
#endif
    bool fast = false;
    bo_config cfg;
    int device;
    std::vector<void *> allocs;
    int *d_go = nullptr, *d_action = nullptr;
    std::vector<int> h_i32;  // scratch [G]
    std::vector<HostRng> rng;          // one legacy MT19937 stream per game slot (bo_rng_seed)
    // The per-move host exchange goes through two device blocks and their pinned mirrors, one copy each:
    //   res  = [res_n | res_best_idx | res_best_mv | res_total | res_idx[RES_CAP] | res_val[RES_CAP]]  (G rows each)
    //   info = [phase | req_node | root_nlegal | root_term | ply]
    int *d_res_blk = nullptr, *d_info_blk = nullptr;
    int *h_res = nullptr, *h_info = nullptr;  // pinned
    int watch_seen = 0;                        // OR of the watched status word over the fetched result blocks (bo_engine_watch)
    double *h_noise = nullptr;                // pinned [G][256]
    int *h_go = nullptr;                      // pinned [G]
    int *h_action = nullptr;                  // pinned [2][G]: the actions of two consecutive bo_play calls (read by the kernel itself)
    int action_flip = 0;
    rt_event ev_action[2]{};                  // behind the bo_k_play that reads each half: a third bo_play waits for the first one's kernel
    bool ev_action_made[2] = {false, false};
    bool ship = true;                         // the turn's small blocks move by bo_k_ship instead of copy commands (BETAONE_TURN_COPIES=1: copies)
    bool prefetch_valid = false;              // bo_search_result_prefetch has been enqueued behind the searches and nothing was stepped since
    std::vector<int> h_nl, h_term;
    bool nl_valid = false;  // h_nl holds the current roots' legal-move counts (set by bo_selfplay_begin)
    void *setup_dev = nullptr, *setup_host = nullptr;  // staging of bo_games_reset_ex (grow-only)
    size_t setup_cap = 0;
    std::vector<unsigned char> noise_pending;  // roots begun by bo_selfplay_turn(defer_noise) whose Dirichlet draw is still due
    // bo_selfplay_turn(flag 4): searches begun on the device; the roots' state is on its way (ev_begin), bo_selfplay_begun collects it
    std::vector<int> lazy_want;
    bool begin_lazy = false, ev_begin_made = false;
    rt_event ev_begin{};
    // bo_selfplay_autoturn: the turn on the device.  In: the choice() uniforms and per-game flags (pinned, read by the kernel itself), the
    // temperature table; out (behind the five info rows, one bo_k_ship): action | state | compact result [G][8]
    double *h_turn_u = nullptr;
    int *h_turn_flags = nullptr;
    double *d_turn_pw = nullptr;
    std::vector<double> turn_pw_host;
    double turn_pw_tfinal = 0.0;  // the T_final the table was built for (0: none yet)
    int *d_turn_action = nullptr, *d_turn_state = nullptr, *d_turn_cres = nullptr;
    std::vector<int> turn_want;
    bool turn_outstanding = false, ev_turn_made = false;
    rt_event ev_turn{};
    template <class T> int alloc(T **p, size_t n) {
        void *v = nullptr;
        int rc = rt_malloc(&v, n * sizeof(T));
        if (rc) return rc;
        allocs.push_back(v);
        *p = (T *)v;
        return 0;
    }
};

extern "C" int bo_abi_version(void) { return BO_ABI_VERSION; }
extern "C" const char *bo_last_error(void) { return g_err.c_str(); }

// ---- text -> plain data (host) ---------------------------------------------------------------------
static const char *START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1";

static bool parse_fen(const char *fen, DPos *out) {
    memset(out, 0, sizeof(*out));
    const char *s = fen;
    while (*s == ' ') s++;
    int r = 7, f = 0;
    for (; *s && *s != ' '; s++) {
        char c = *s;
        if (c == '/') { r--; f = 0; continue; }
        if (c >= '1' && c <= '8') { f += c - '0'; continue; }
        int white = (c >= 'A' && c <= 'Z');
        char l = (char)(white ? c - 'A' + 'a' : c);
        int t = l == 'p' ? 0 : l == 'n' ? 1 : l == 'b' ? 2 : l == 'r' ? 3 : l == 'q' ? 4 : l == 'k' ? 5 : -1;
        if (t < 0 || r < 0 || f > 7) return false;
        out->bb[t] |= BIT(r * 8 + f);
        out->bb[white ? BB_WHITE : BB_BLACK] |= BIT(r * 8 + f);
        f++;
    }
    int turn = 1, ep = -1, half = 0, full = 1;
    uint32_t cr = 0;
    while (*s == ' ') s++;
    if (*s) { if (*s == 'w') turn = 1; else if (*s == 'b') turn = 0; else return false; s++; }
    while (*s == ' ') s++;
    for (; *s && *s != ' '; s++) {
        if (*s == 'K') cr |= 0x02u; else if (*s == 'Q') cr |= 0x04u; else if (*s == 'k') cr |= 0x08u;
        else if (*s == 'q') cr |= 0x10u; else if (*s != '-') return false;
    }
    // python-chess clean_castling_rights(): rook on its corner, king on e1/e8
    const uint64_t wk = out->bb[BB_K] & out->bb[BB_WHITE] & BIT(4), bk = out->bb[BB_K] & out->bb[BB_BLACK] & BIT(60);
    const uint64_t wr = out->bb[BB_R] & out->bb[BB_WHITE], br = out->bb[BB_R] & out->bb[BB_BLACK];
    if (!wk || !(wr & BIT(7))) cr &= ~0x02u;
    if (!wk || !(wr & BIT(0))) cr &= ~0x04u;
    if (!bk || !(br & BIT(63))) cr &= ~0x08u;
    if (!bk || !(br & BIT(56))) cr &= ~0x10u;
    while (*s == ' ') s++;
    if (*s) {
        if (*s != '-') {
            if (s[0] < 'a' || s[0] > 'h' || s[1] < '1' || s[1] > '8') return false;
            ep = (s[1] - '1') * 8 + (s[0] - 'a');
            s += 2;
        } else s++;
    }
    while (*s == ' ') s++;
    if (*s) half = (int)strtol(s, (char **)&s, 10);
    while (*s == ' ') s++;
    if (*s) full = (int)strtol(s, (char **)&s, 10);
    if (full < 1) full = 1;
    out->flags = (turn ? F_TURN : 0u) | cr | ((uint32_t)(ep + 1) << F_EP_SHIFT);
    out->halfmove = half;
    out->fullmove = full;
    return true;
}

static bool parse_uci_moves(const char *s, std::vector<bo_mv> *out) {
    out->clear();
    if (!s) return true;
    while (*s) {
        while (*s == ' ') s++;
        if (!*s) break;
        const char *b = s;
        while (*s && *s != ' ') s++;
        size_t n = (size_t)(s - b);
        if (n < 4 || n > 5) return false;
        if (b[0] < 'a' || b[0] > 'h' || b[2] < 'a' || b[2] > 'h' || b[1] < '1' || b[1] > '8' || b[3] < '1' || b[3] > '8') return false;
        int from = (b[1] - '1') * 8 + (b[0] - 'a'), to = (b[3] - '1') * 8 + (b[2] - 'a'), promo = 0;
        if (n == 5) {
            promo = b[4] == 'n' ? 2 : b[4] == 'b' ? 3 : b[4] == 'r' ? 4 : b[4] == 'q' ? 5 : -1;
            if (promo < 0) return false;
        }
        out->push_back(MV(from, to, promo));
    }
    return true;
}

static DPos from_abi(const bo_position &p) {
    DPos d;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < 8; i++) d.bb[i] = p.bb[i];
    d.flags = (p.turn ? F_TURN : 0u) | ((p.castling & 0xFu) << F_CASTLE_SHIFT) | ((uint32_t)(p.ep_square + 1) << F_EP_SHIFT);
    if (p.ep_key >= 0) d.flags |= (uint32_t)(p.ep_key + 1) << F_EPKEY_SHIFT;
    // ep_key == -1 ("no ep component in the key") with a raw ep square present is re-derived on the device by
    // finish_key(); that is deterministic and gives -1 again.  Tracker keys taken from python-chess key tuples
    // carry ep_square == ep_key.
    d.halfmove = p.halfmove_clock;
    d.fullmove = p.fullmove_number;
    return d;
}
static void to_abi(const DPos &d, bo_position *p) {
    for (int i = 0; i < 8; i++) p->bb[i] = d.bb[i];
    p->turn = (int)(d.flags & F_TURN);
    p->castling = (d.flags & F_CASTLE_MASK) >> F_CASTLE_SHIFT;
    p->ep_square = (int)((d.flags & F_EP_MASK) >> F_EP_SHIFT) - 1;
    p->ep_key = (int)((d.flags & F_EPKEY_MASK) >> F_EPKEY_SHIFT) - 1;
    p->halfmove_clock = d.halfmove;
    p->fullmove_number = d.fullmove;
}

// ---- create / destroy -------------------------------------------------------------------------------
extern "C" int bo_engine_create(const bo_config *cfg, int device, bo_engine **out) {
    if (!cfg || !out) return fail(BO_E_ARG, "null argument");
    if (cfg->n_games < 1 || cfg->num_simulations < 0 || cfg->mcts_batch_size < 1 || cfg->max_plies < 2)
        return fail(BO_E_ARG, "n_games/num_simulations/mcts_batch_size/max_plies out of range");
    const bool fast = cfg->mode == 1;
    if (cfg->mode != 0 && cfg->mode != 1) return fail(BO_E_ARG, "mode must be 0 (reference semantics) or 1 (fast)");
    if (fast && (cfg->leaves_per_step < 1 || cfg->leaves_per_step > BO_FW_LMAX)) return fail(BO_E_ARG, "leaves_per_step out of range (1..64)");
    const int root_m = fast ? 1 : (int)(cfg->widen_coeff * sqrt(1.0));
    const int ch_max = fast ? 1 : (int)(cfg->widen_coeff * sqrt((double)cfg->mcts_batch_size));
    if (!fast && (cfg->widen_coeff < 1.0 || ch_max > BO_CH_CAP))
        return fail(BO_E_CONFIG, "WIDEN_COEFF must be >= 1 and int(WIDEN_COEFF*sqrt(MCTS_BATCH_SIZE)) <= 32");
    RT(rt_set_device(device));
    bo_engine *e = new bo_engine();
    e->cfg = *cfg;
    e->device = device;
    EngCfg &c = e->d.c;
    c.G = cfg->n_games; c.S = cfg->num_simulations; c.B = cfg->mcts_batch_size;
    e->fast = fast;
    // Node capacity in reference semantics: a leaf evaluated with k rows of a batch gets int(W*sqrt(k)) children (mcts.py:55-57), and
    // the k over all evaluated leaves of a search sum to at most S, so at most S * max_k int(W*sqrt(k))/k nodes beside the root's
    // (= S for the reference's W = 1.5; a larger WIDEN_COEFF creates more nodes per simulation).
    double per_sim = 1.0;
    for (int k = 1; k <= cfg->mcts_batch_size; k++) {
        const double r = (double)(int)(cfg->widen_coeff * sqrt((double)k)) / (double)k;
        if (r > per_sim) per_sim = r;
    }
    // (fast mode keeps its trees in child-block arenas, bo_fastw.h; the node arrays below then only hold the root's slot)
    c.NCAP = fast ? 2 : (int)ceil(c.S * per_sim) + 2 * root_m + 4;
    c.nstride = c.NCAP;
    c.PLY_CAP = cfg->max_plies; c.TRK_CAP = cfg->max_plies;
    c.UL_MAX = fast ? 1 : c.B; c.CH_MAX = ch_max < 1 ? 1 : ch_max;
    c.cpuct = (float)cfg->cpuct;
    c.keep = (float)(1.0 - cfg->dirichlet_epsilon);
    c.eps = cfg->dirichlet_epsilon;
    c.use_noise = cfg->dirichlet_alpha > 0 ? 1 : 0;
    c.root_m = root_m;
    c.profile = 0;
    { const char *v = getenv("BETAONE_BURST_TWO_PATHS"); c.burst_two = (v && v[0] == '0') ? 0 : 1; }
    const size_t G = (size_t)c.G, N = G * (size_t)c.NCAP;
    Eng &d = e->d;
    int rc = 0;
    float *sq = nullptr, *rcp = nullptr; int *wl = nullptr;
    rc |= e->alloc(&sq, (size_t)c.S + 2); rc |= e->alloc(&wl, (size_t)c.B + 1); rc |= e->alloc(&rcp, (size_t)c.S + 3);
    int **iscal[] = {&d.sims_done, &d.n_nodes, &d.rows, &d.n_runs, &d.n_ul, &d.req_nlegal, &d.status,
                     &d.trk_n, &d.n_hist, &d.ctx_mode, &d.root_nch, &d.stat_evals,
                     &d.stat_flushes, &d.stat_term_sims, &d.stat_levels, &d.stat_children_scanned, &e->d_go, &e->d_action};
    for (int **p : iscal) rc |= e->alloc(p, G);
    const size_t res_ints = G * (4 + 2 * (size_t)BO_RES_CAP) + 4, info_ints = G * 15;  // (+ 4: the watched status word, bo_engine_watch; info rows 5..14: bo_selfplay_autoturn's outputs)
    rc |= e->alloc(&e->d_res_blk, res_ints); rc |= e->alloc(&e->d_info_blk, info_ints);
    if (!rc) {
        d.res_n = e->d_res_blk; d.res_best_idx = e->d_res_blk + G; d.res_best_mv = e->d_res_blk + 2 * G; d.res_total = e->d_res_blk + 3 * G;
        d.res_idx = e->d_res_blk + 4 * G; d.res_val = reinterpret_cast<float *>(e->d_res_blk + 4 * G + G * BO_RES_CAP);
        d.res_watch = e->d_res_blk + G * (4 + 2 * (size_t)BO_RES_CAP); d.watch = nullptr; d.watch_n = 1;
        d.phase = e->d_info_blk; d.req_node = e->d_info_blk + G; d.root_nlegal = e->d_info_blk + 2 * G; d.root_term = e->d_info_blk + 3 * G;
        d.ply = e->d_info_blk + 4 * G;
        e->d_turn_action = e->d_info_blk + 5 * G; e->d_turn_state = e->d_info_blk + 6 * G; e->d_turn_cres = e->d_info_blk + 7 * G;
        rt_memset(e->d_res_blk, 0, res_ints * 4, nullptr); rt_memset(e->d_info_blk, 0, info_ints * 4, nullptr);
    }
    rc |= rt_host_alloc((void **)&e->h_res, res_ints * 4); rc |= rt_host_alloc((void **)&e->h_info, info_ints * 4);
    rc |= rt_host_alloc((void **)&e->h_noise, G * BO_MAX_MOVES * sizeof(double)); rc |= rt_host_alloc((void **)&e->h_go, G * 4);
    rc |= rt_host_alloc((void **)&e->h_action, 2 * G * 4);
    rc |= rt_host_alloc((void **)&e->h_turn_u, G * sizeof(double)); rc |= rt_host_alloc((void **)&e->h_turn_flags, G * 4);
    rc |= e->alloc(&e->d_turn_pw, (size_t)cfg->num_simulations + 1);
    { const char *v = getenv("BETAONE_TURN_COPIES"); e->ship = !(v && v[0] == '1'); }
    rc |= e->alloc(&d.gpos, G * c.PLY_CAP); rc |= e->alloc(&d.trk, G * c.TRK_CAP); rc |= e->alloc(&d.trk_cnt, G * c.TRK_CAP);
    rc |= e->alloc(&d.hist, G * 7);
    rc |= e->alloc(&d.n_visits, N); rc |= e->alloc(&d.parent, N); rc |= e->alloc(&d.first_child, N); rc |= e->alloc(&d.n_children, N);
    rc |= e->alloc(&d.q, N); rc |= e->alloc(&d.prior, N); rc |= e->alloc(&d.move, N); rc |= e->alloc(&d.term, N);
    rc |= e->alloc(&d.eval_slot, N); rc |= e->alloc(&d.npos, N);
    rc |= e->alloc(&d.run_leaf, G * c.B); rc |= e->alloc(&d.run_cnt, G * c.B);
    rc |= e->alloc(&d.ul_node, G * c.UL_MAX); rc |= e->alloc(&d.ul_nlegal, G * c.UL_MAX); rc |= e->alloc(&d.ul_value, G * c.UL_MAX);
    rc |= e->alloc(&d.ul_move, G * c.UL_MAX * BO_CH_CAP); rc |= e->alloc(&d.ul_prior, G * c.UL_MAX * BO_CH_CAP);
    rc |= e->alloc(&d.req_moves, G * BO_MAX_MOVES); rc |= e->alloc(&d.root_moves, G * BO_MAX_MOVES);
    rc |= e->alloc(&d.root_child_rank, G * 2 * BO_CH_CAP); rc |= e->alloc(&d.noise, G * BO_MAX_MOVES);
    rc |= e->alloc(&d.played, G * c.PLY_CAP);
    rc |= e->alloc(&d.prof, G * BO_PROF_SLOTS);
    d.played_now = nullptr;
    if (fast) {
        FastW &f = e->f;
        const size_t L = (size_t)cfg->leaves_per_step;
        f.L = (int)L;
        f.sel_ut = 2; f.sel_flags = FW_SEL_OCT;  // measured fastest on an MI355X (profiles/r03_fast_select_variants.md): eight lanes per game
                                                 // (more than 8 leaves per step: the half-wave forms)
        // Arena capacity per game, in granules of BO_FW_GR records.  A search creates at most S + L + 1 runs (one per expanded
        // node: header + 1..32 record granules, ~6 granules = 768 bytes at chess's ~35 legal moves) on top of the subtree kept
        // from the previous search: room for 24 granules (3 KB) per possible expansion of this and of the previous search.
        // A run that still does not fit is refused (status bit) -- the search goes on with that leaf unexpanded.
        const size_t expansions = (size_t)c.S + L + 2;
        f.NG = cfg->fast_arena_granules > 0 ? cfg->fast_arena_granules : (int)((6 * expansions + 8) * (32 / BO_FW_GR));
        if (f.NG < 64 || f.NG > BO_FW_LINK_MASK) { bo_engine_destroy(e); return fail(BO_E_ARG, "fast_arena_granules out of range (64 .. 2^24 - 1)"); }
        const size_t NR = (size_t)f.NG * BO_FW_GR;
        rc |= e->alloc(&f.arena, G * 2 * NR); rc |= e->alloc(&f.amove, G * 2 * NR);
        f.CS = (FWC_HEAD + FWR_FIELDS * (int)L + 31) / 32 * 32;
        rc |= e->alloc(&f.ctl, G * (size_t)f.CS);
        rc |= e->alloc(&f.played_now, G); rc |= e->alloc(&f.row_pos, G * L);
        rc |= e->alloc(&f.row_moves, G * L * BO_MAX_MOVES);
        rc |= e->alloc(&f.sim_path, G * L * BO_FW_PATH_CAP);
        float *st = nullptr, *rt = nullptr;
        rc |= e->alloc(&st, (size_t)BO_FW_SQRT_TAB); rc |= e->alloc(&rt, (size_t)BO_FW_RCP_TAB);
        if (!rc) {
            // RN(sqrt(n)) and RN(1 / k): the double result rounded once more is the correctly rounded binary32 one (53 >= 2 * 24 + 2)
            std::vector<float> hs2(BO_FW_SQRT_TAB), hr2(BO_FW_RCP_TAB, 0.0f);
            for (int n = 0; n < BO_FW_SQRT_TAB; n++) hs2[n] = (float)sqrt((double)n);
            for (int k = 1; k < BO_FW_RCP_TAB; k++) hr2[k] = (float)(1.0 / (double)k);
            rt_h2d(st, hs2.data(), hs2.size() * sizeof(float), nullptr); rt_h2d(rt, hr2.data(), hr2.size() * sizeof(float), nullptr);
            rt_sync(nullptr);
            f.sqrt_tab = st; f.rcp_tab = rt;
            rt_memset(f.ctl, 0, G * (size_t)f.CS * 4, nullptr);
            rt_memset(f.played_now, 0, G * 4, nullptr);
            d.played_now = f.played_now;
        }
    }
    if (rc) { bo_engine_destroy(e); return fail(BO_E_HIP, "device allocation failed"); }
    // host-built lookup tables: Python's math.sqrt / int() in double, rounded to binary32 once
    std::vector<float> hs((size_t)c.S + 2);
    for (int n = 0; n < c.S + 2; n++) hs[n] = (float)sqrt((double)n + 1e-8);  // mcts.py:93
    std::vector<int> hw((size_t)c.B + 1);
    for (int k = 0; k <= c.B; k++) hw[k] = (int)(cfg->widen_coeff * sqrt((double)k));  // mcts.py:55-57 with n_visits+1 == k
    std::vector<float> hr((size_t)c.S + 3, 0.0f);
    for (int n = 1; n < c.S + 3; n++) hr[n] = (float)(1.0 / (double)n);  // RN(1/n): rounding the double quotient again is innocuous (53 >= 2*24+2)
    rt_h2d(rcp, hr.data(), hr.size() * sizeof(float), nullptr);
    d.rcp_lut = rcp;
    rt_h2d(sq, hs.data(), hs.size() * sizeof(float), nullptr);
    rt_h2d(wl, hw.data(), hw.size() * sizeof(int), nullptr);
    d.sqrt_lut = sq; d.widen_lut = wl;
    for (int **p : iscal) rt_memset(*p, 0, G * sizeof(int), nullptr);
    rt_memset(d.noise, 0, G * BO_MAX_MOVES * sizeof(double), nullptr);
    rt_sync(nullptr);
    e->h_i32.resize(G);
    e->rng.resize(G);
    for (size_t g = 0; g < G; g++) hr_seed(&e->rng[g], (uint32_t)g);
    memset(e->h_noise, 0, G * BO_MAX_MOVES * sizeof(double));
    e->h_nl.resize(G); e->h_term.resize(G);
    *out = e;
    return BO_OK;
}

extern "C" void bo_engine_destroy(bo_engine *e) {
    if (!e) return;
    rt_set_device(e->device);
    for (void *p : e->allocs) rt_free(p);
    rt_host_free(e->h_res); rt_host_free(e->h_info); rt_host_free(e->h_noise); rt_host_free(e->h_go); rt_host_free(e->h_action);
    rt_host_free(e->h_turn_u); rt_host_free(e->h_turn_flags);
    if (e->ev_turn_made) rt_event_destroy(e->ev_turn);
    for (int i = 0; i < 2; i++) if (e->ev_action_made[i]) rt_event_destroy(e->ev_action[i]);
    if (e->setup_dev) rt_free(e->setup_dev);
    if (e->setup_host) rt_host_free(e->setup_host);
    if (e->ev_begin_made) rt_event_destroy(e->ev_begin);
#if !defined(BO_WAVE_EMU)
    if (e->sel_ev0) { (void)hipEventDestroy(e->sel_ev0); (void)hipEventDestroy(e->sel_ev1); }
#endif
    delete e;
}

// ---- set-up ---------------------------------------------------------------------------------------------
extern "C" int bo_games_reset_ex(bo_engine *e, int n, const int32_t *slots, const char *const *fens, const char *const *moves,
                                 const bo_position *hist, const int32_t *n_hist, const bo_position *trk,
                                 const int32_t *trk_counts, const int32_t *trk_off, void *stream) {
    if (!e || n < 1 || !slots) return fail(BO_E_ARG, "bad arguments");
    const EngCfg &c = e->d.c;
    std::vector<DPos> start((size_t)n);
    std::vector<std::vector<bo_mv>> mv((size_t)n);
    size_t max_moves = 1, max_trk = 1;
    for (int i = 0; i < n; i++) {
        if (slots[i] < 0 || slots[i] >= c.G) return fail(BO_E_ARG, "slot out of range");
        if (!parse_fen(fens && fens[i] ? fens[i] : START_FEN, &start[i])) return fail(BO_E_FEN, "bad FEN");
        if (!parse_uci_moves(moves ? moves[i] : nullptr, &mv[i])) return fail(BO_E_FEN, "bad UCI move list");
        if ((int)mv[i].size() + 2 > c.PLY_CAP) return fail(BO_E_ARG, "move list longer than max_plies");
        if (mv[i].size() > max_moves) max_moves = mv[i].size();
        if (trk_off && (size_t)(trk_off[i + 1] - trk_off[i]) > max_trk) max_trk = (size_t)(trk_off[i + 1] - trk_off[i]);
    }
    if (max_trk > (size_t)c.TRK_CAP) return fail(BO_E_ARG, "tracker larger than max_plies");
    std::vector<bo_mv> flat((size_t)n * max_moves, 0);
    std::vector<int> nm((size_t)n), nh((size_t)n, -1), nt((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        nm[i] = (int)mv[i].size();
        for (size_t k = 0; k < mv[i].size(); k++) flat[(size_t)i * max_moves + k] = mv[i][k];
    }
    std::vector<DPos> hh, tt;
    std::vector<int> tc;
    if (n_hist) {
        hh.resize((size_t)n * 7);
        tt.resize((size_t)n * max_trk);
        tc.resize((size_t)n * max_trk, 0);
        for (int i = 0; i < n; i++) {
            nh[i] = n_hist[i];
            if (nh[i] > 7) return fail(BO_E_ARG, "more than 7 history boards");
            for (int k = 0; k < nh[i]; k++) hh[(size_t)i * 7 + k] = from_abi(hist[(size_t)i * 7 + k]);
            nt[i] = trk_off ? trk_off[i + 1] - trk_off[i] : 0;
            for (int k = 0; k < nt[i]; k++) {
                tt[(size_t)i * max_trk + k] = from_abi(trk[trk_off[i] + k]);
                tc[(size_t)i * max_trk + k] = trk_counts[trk_off[i] + k];
            }
        }
    }
    // One packed upload through the engine's persistent staging buffers (grow-only device + pinned host memory): refilling
    // a finished game's slot happens about once per ply in steady-state self-play, and nine hipMalloc / hipFree pairs per
    // call cost more than the set-up kernel itself.
    std::vector<int> sl(slots, slots + n);
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return o; };
    const size_t o_slots = place(sl.size() * 4), o_start = place(start.size() * sizeof(DPos)), o_mv = place(flat.size() * sizeof(bo_mv)),
                 o_nm = place(nm.size() * 4);
    size_t o_nh = 0, o_hh = 0, o_tt = 0, o_tc = 0, o_nt = 0;
    if (n_hist) {
        o_nh = place(nh.size() * 4); o_hh = place(hh.size() * sizeof(DPos)); o_tt = place(tt.size() * sizeof(DPos));
        o_tc = place(tc.size() * 4); o_nt = place(nt.size() * 4);
    }
    int rc = 0;
    if (off > e->setup_cap) {
        if (e->setup_dev) rt_free(e->setup_dev);
        if (e->setup_host) rt_host_free(e->setup_host);
        e->setup_dev = nullptr; e->setup_host = nullptr;
        e->setup_cap = 2 * off + (256u << 10);  // roomy from the start: regrowing means hipFree, i.e. a device synchronisation
        rc = rt_malloc(&e->setup_dev, e->setup_cap) | rt_host_alloc(&e->setup_host, e->setup_cap);
        if (rc) e->setup_cap = 0;
    }
    SetupArgs a;
    memset(&a, 0, sizeof(a));
    if (!rc) {
        char *h = (char *)e->setup_host, *dv = (char *)e->setup_dev;
        memcpy(h + o_slots, sl.data(), sl.size() * 4); memcpy(h + o_start, start.data(), start.size() * sizeof(DPos));
        memcpy(h + o_mv, flat.data(), flat.size() * sizeof(bo_mv)); memcpy(h + o_nm, nm.data(), nm.size() * 4);
        a.slots = (const int *)(dv + o_slots); a.start = (const DPos *)(dv + o_start); a.moves = (const bo_mv *)(dv + o_mv);
        a.n_moves = (const int *)(dv + o_nm); a.max_moves = (int)max_moves; a.max_trk = (int)max_trk;
        if (n_hist) {
            memcpy(h + o_nh, nh.data(), nh.size() * 4); memcpy(h + o_hh, hh.data(), hh.size() * sizeof(DPos));
            memcpy(h + o_tt, tt.data(), tt.size() * sizeof(DPos)); memcpy(h + o_tc, tc.data(), tc.size() * 4);
            memcpy(h + o_nt, nt.data(), nt.size() * 4);
            a.n_hist = (const int *)(dv + o_nh); a.hist = (const DPos *)(dv + o_hh); a.trk = (const DPos *)(dv + o_tt);
            a.trk_cnt = (const int *)(dv + o_tc); a.n_trk = (const int *)(dv + o_nt);
        }
        rc = rt_h2d(dv, h, off, stream);
        if (!rc) rc = RT_LAUNCH(bo_k_setup, n, stream, e->d, a);
        if (!rc && e->fast) rc = RT_LAUNCH(bo_k_fw_reset, n, stream, e->d, e->f, a.slots);
    }
    int rc2 = rt_sync(stream);  // the staging buffers are free again when this returns
    if (rc || rc2) return fail(BO_E_HIP, std::string("bo_games_reset: ") + rt_errstr(rc ? rc : rc2));
    return BO_OK;
}

extern "C" int bo_games_reset(bo_engine *e, int n, const int32_t *slots, const char *const *fens, const char *const *moves,
                              void *stream) {
    return bo_games_reset_ex(e, n, slots, fens, moves, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int bo_root_info(bo_engine *e, int32_t *n_legal, int32_t *terminal, int32_t *ply, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    const size_t G = (size_t)e->d.c.G;
    RT(rt_d2h(e->h_info + 2 * G, e->d_info_blk + 2 * G, 3 * G * 4, stream));  // [root_nlegal | root_term | ply] in one copy
    RT(rt_sync(stream));
    if (n_legal) memcpy(n_legal, e->h_info + 2 * G, G * 4);
    if (terminal) memcpy(terminal, e->h_info + 3 * G, G * 4);
    if (ply) memcpy(ply, e->h_info + 4 * G, G * 4);
    return BO_OK;
}

// ---- search --------------------------------------------------------------------------------------------
static int upload_noise(bo_engine *e, const unsigned char *rows_used, void *stream) {
    const size_t G = (size_t)e->d.c.G;
    size_t cols = 1;  // only the columns some root uses travel (apply_root reads noise[g][j < n_legal(g)])
    for (size_t g = 0; g < G; g++)
        if (rows_used[g] && (size_t)e->h_nl[g] > cols) cols = (size_t)e->h_nl[g];
    if (e->nl_valid && cols < BO_MAX_MOVES)
        RT(rt_h2d_2d(e->d.noise, BO_MAX_MOVES * sizeof(double), e->h_noise, BO_MAX_MOVES * sizeof(double), cols * sizeof(double), G, stream));
    else
        RT(rt_h2d(e->d.noise, e->h_noise, G * BO_MAX_MOVES * sizeof(double), stream));
    return BO_OK;
}

// noise_later: the caller uploads the noise before the first bo_step that applies a root evaluation (bo_selfplay_noise)
static int search_begin_impl(bo_engine *e, const int32_t *go, const double *noise, bool noise_later, float *nn_in_dev, void *stream) {
    if (!e || !go || !nn_in_dev) return fail(BO_E_ARG, "null argument");
    const size_t G = (size_t)e->d.c.G;
    if (e->d.c.use_noise && !noise && !noise_later) return fail(BO_E_ARG, "noise required when dirichlet_alpha > 0");
    if (go != e->h_go) memcpy(e->h_go, go, G * 4);  // pinned staging: one DMA, no pageable bounce buffer
    RT(rt_h2d(e->d_go, e->h_go, G * 4, stream));
    if (noise) {
        if (noise != e->h_noise) memcpy(e->h_noise, noise, G * BO_MAX_MOVES * sizeof(double));
        std::vector<unsigned char> used(G);
        for (size_t g = 0; g < G; g++) used[g] = e->h_go[g] != 0;
        int rc = upload_noise(e, used.data(), stream);
        if (rc) return rc;
    }
    e->nl_valid = false;
    if (e->fast) RT(RT_LAUNCH(bo_k_fw_search_begin, e->d.c.G, stream, e->d, e->f, (const int *)e->d_go, nn_in_dev));
    else RT(RT_LAUNCH(bo_k_search_begin, e->d.c.G, stream, e->d, (const int *)e->d_go, nn_in_dev));
    return BO_OK;
}

extern "C" int bo_search_begin(bo_engine *e, const int32_t *go, const double *noise, float *nn_in_dev, void *stream) {
    return search_begin_impl(e, go, noise, false, nn_in_dev, stream);
}

// bo_k_fw_select comes in one instantiation per (games per half-wave, leaves-per-step capacity, FW_SEL_* flags)
static int launch_fw_select(bo_engine *e, const float *value_dev, int kind, void *stream) {
    const FastW &f = e->f;
    const int G = e->d.c.G, L = f.L;
    if (L == 4 && (f.sel_flags & FW_SEL_LANE)) {  // one lane per game (bo_fastw.h)
        const int blocks = (G + 63) / 64;
        if (f.sel_flags & FW_SEL_NT) return RT_LAUNCH(bo_k_fw_select_lane_nt, blocks, stream, e->d, e->f, value_dev, kind);
        return RT_LAUNCH(bo_k_fw_select_lane, blocks, stream, e->d, e->f, value_dev, kind);
    }
    if (L <= 8 && (f.sel_flags & FW_SEL_QUAD)) {  // four lanes per game, sixteen games per wave
        const int blocks = (G + 15) / 16;
#define SELQ(K) return RT_LAUNCH(K, blocks, stream, e->d, e->f, value_dev, kind)
        if (L > 4) SELQ(bo_k_fw_select_q4l8_0);
        if (f.sel_flags & FW_SEL_DENSE) SELQ(bo_k_fw_select_q4l4_4);
        if (f.sel_flags & FW_SEL_ROOT_IN_REGS) SELQ(bo_k_fw_select_q4l4_2);
        SELQ(bo_k_fw_select_q4l4_0);
#undef SELQ
    }
    if (L <= 8 && (f.sel_flags & FW_SEL_OCT)) {  // eight lanes per game, eight games per wave
        const int blocks = (G + 7) / 8;
        const int fl = f.sel_flags & 3;
#define SELO(K) return RT_LAUNCH(K, blocks, stream, e->d, e->f, value_dev, kind)
        if (L > 4) { if (fl & 2) SELO(bo_k_fw_select_o8l8_2); SELO(bo_k_fw_select_o8l8_0); }
        if (fl == 3) SELO(bo_k_fw_select_o8l4_3);
        if (fl == 2) SELO(bo_k_fw_select_o8l4_2);
        if (fl == 1) SELO(bo_k_fw_select_o8l4_1);
        SELO(bo_k_fw_select_o8l4_0);
#undef SELO
    }
    const int ut = L > 16 ? 1 : (L > 8 && f.sel_ut > 2) ? 2 : L > 4 ? (L > 8 ? 2 : 4) : f.sel_ut;
    const int grid = (G + 2 * ut - 1) / (2 * ut);
#define SEL(K) return RT_LAUNCH(K, grid, stream, e->d, e->f, value_dev, kind)
    const bool fancy = (f.sel_flags & FW_SEL_ROOT_IN_REGS) != 0;  // (L > 4: plain, or non-temporal + root in registers)
    if (L > 16) { if (fancy) SEL(bo_k_fw_select_u1l64_3); SEL(bo_k_fw_select_u1l64_0); }
    if (L > 8) { if (fancy) SEL(bo_k_fw_select_u2l16_3); SEL(bo_k_fw_select_u2l16_0); }
    if (L > 4) { if (fancy) SEL(bo_k_fw_select_u4l8_3); SEL(bo_k_fw_select_u4l8_0); }
#define SEL8(UT)                                                                                  \
    switch (f.sel_flags & 7) {                                                                     \
        case 0: SEL(bo_k_fw_select_u##UT##l4_0); case 1: SEL(bo_k_fw_select_u##UT##l4_1);         \
        case 2: SEL(bo_k_fw_select_u##UT##l4_2); case 3: SEL(bo_k_fw_select_u##UT##l4_3);         \
        case 4: SEL(bo_k_fw_select_u##UT##l4_4); case 5: SEL(bo_k_fw_select_u##UT##l4_5);         \
        case 6: SEL(bo_k_fw_select_u##UT##l4_6); default: SEL(bo_k_fw_select_u##UT##l4_7);        \
    }
    if (ut >= 4) { SEL8(4) }
    SEL8(2)
#undef SEL8
#undef SEL
}

static int step_launch(bo_engine *e, const float *policy_dev, const float *value_dev, int policy_kind, float *nn_in_dev, void *stream, const StepTail &vt) {
    e->prefetch_valid = false;  // (a prefetched result block is older than this step; replays of a captured step are the caller's to track)
    if (e->fast) {  // apply (one wave per row) -> backup + select (half a wave per game) -> leaf positions and planes (one wave per row)
        const int rows = e->d.c.G * e->f.L;
        if (policy_kind != BO_POLICY_NONE) RT(RT_LAUNCH(bo_k_fw_apply, rows, stream, e->d, e->f, policy_dev, policy_kind));
#if !defined(BO_WAVE_EMU)
        if (e->sel_profile) {  // eager launches only (events recorded during a graph capture would become graph nodes)
            if (!e->sel_ev0) { RT((int)hipEventCreate(&e->sel_ev0)); RT((int)hipEventCreate(&e->sel_ev1)); }
            if (e->sel_pending) {
                float ms = 0.0f;
                RT((int)hipEventSynchronize(e->sel_ev1));
                RT((int)hipEventElapsedTime(&ms, e->sel_ev0, e->sel_ev1));
                e->sel_ms += ms; e->sel_launches++; e->sel_pending = 0;
            }
            RT((int)hipEventRecord(e->sel_ev0, (hipStream_t)stream));
        }
#endif
        RT(launch_fw_select(e, value_dev, policy_kind, stream));
#if !defined(BO_WAVE_EMU)
        if (e->sel_profile) { RT((int)hipEventRecord(e->sel_ev1, (hipStream_t)stream)); e->sel_pending = 1; }
#endif
        RT(RT_LAUNCH(bo_k_fw_leaf, rows, stream, e->d, e->f, nn_in_dev));
    } else {
        RT(RT_LAUNCH(bo_k_step, e->d.c.G, stream, e->d, policy_dev, value_dev, policy_kind, nn_in_dev, vt));
    }
    return BO_OK;
}
extern "C" int bo_step(bo_engine *e, const float *policy_dev, const float *value_dev, int policy_kind, float *nn_in_dev,
                       void *stream) {
    if (!e || !nn_in_dev) return fail(BO_E_ARG, "null argument");
    if (policy_kind != BO_POLICY_NONE && (!policy_dev || !value_dev)) return fail(BO_E_ARG, "policy/value required");
    return step_launch(e, policy_dev, value_dev, policy_kind, nn_in_dev, stream, StepTail{nullptr, nullptr, nullptr, nullptr, 0});
}
extern "C" int bo_step_heads(bo_engine *e, const float *logits_dev, const float *vpart_dev, const float *b1_dev, const float *w2_dev,
                             const float *b2_dev, int rows, float *nn_in_dev, void *stream) {
    if (!e || !nn_in_dev || !logits_dev || !vpart_dev || !b1_dev || !w2_dev || !b2_dev) return fail(BO_E_ARG, "null argument");
    if (e->fast) return fail(BO_E_CONFIG, "bo_step_heads: the reference-semantics search only (fast mode evaluates rows, not games)");
    if (rows < e->d.c.G) return fail(BO_E_ARG, "bo_step_heads: the partial sums must hold a row per game (rows >= G)");
    return step_launch(e, logits_dev, nullptr, BO_POLICY_LOGITS, nn_in_dev, stream, StepTail{vpart_dev, b1_dev, w2_dev, b2_dev, rows});
}

extern "C" int bo_search_poll(bo_engine *e, int32_t *n_running, int32_t *n_requested, int32_t *requested_mask, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    const size_t G = (size_t)e->d.c.G;
    RT(rt_d2h(e->h_info, e->d_info_blk, 2 * G * 4, stream));  // [phase | req_node] in one copy
    const int *ph = e->h_info, *rq = e->h_info + G;
    RT(rt_sync(stream));
    int run = 0, req = 0;
    for (size_t g = 0; g < G; g++) {
        bool r = ph[g] == PH_RUN;
        bool q = r && rq[g] >= 0;
        run += r; req += q;
        if (requested_mask) requested_mask[g] = q ? 1 : 0;
    }
    if (n_running) *n_running = run;
    if (n_requested) *n_requested = req;
    return BO_OK;
}

extern "C" int bo_search_stop(bo_engine *e, const int32_t *stop_mask, int32_t *sims_done, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    if (e->fast) return fail(BO_E_CONFIG, "bo_search_stop: reference-semantics engines only");
    const size_t G = (size_t)e->d.c.G;
    const int *d_mask = nullptr;
    if (stop_mask) {
        memcpy(e->h_go, stop_mask, G * 4);
        RT(rt_h2d(e->d_go, e->h_go, G * 4, stream));
        d_mask = e->d_go;
    }
    RT(RT_LAUNCH(bo_k_stop, e->d.c.G, stream, e->d, d_mask));
    if (sims_done) RT(rt_d2h(sims_done, e->d.sims_done, G * 4, stream));
    RT(rt_sync(stream));
    return BO_OK;
}

// the result kernel and the copy of its block, enqueued (no wait)
static int ship(bo_engine *e, int *dst_a, const int *src_a, size_t n_a, int *dst_b, const int *src_b, size_t n_b, void *stream) {
    const size_t n = n_a > n_b ? n_a : n_b;
    const int blocks = (int)((n + 511) / 512 < 1 ? 1 : ((n + 511) / 512 > 128 ? 128 : (n + 511) / 512));  // ~8 words per lane, at most 128 waves
    RT(RT_LAUNCH(bo_k_ship, blocks, stream, dst_a, src_a, (int)n_a, dst_b, src_b, (int)n_b, blocks));
    (void)e;
    return BO_OK;
}

// with_info: [phase | req_node] of every slot comes along (bo_selfplay_turn's "are all searches finished?")
static int result_enqueue(bo_engine *e, void *stream, bool with_info = false) {
    const size_t G = (size_t)e->d.c.G, res_words = G * (4 + 2 * (size_t)BO_RES_CAP) + 4;  // the whole result block (+ the watched word)
    if (with_info && !e->ship) RT(rt_d2h(e->h_info, e->d_info_blk, 2 * G * 4, stream));
    if (e->fast) RT(RT_LAUNCH(bo_k_fw_result, e->d.c.G, stream, e->d, e->f));
    else RT(RT_LAUNCH(bo_k_result, e->d.c.G, stream, e->d));
    if (e->ship) return ship(e, e->h_res, (const int *)e->d_res_blk, res_words, e->h_info, (const int *)e->d_info_blk, with_info ? 2 * G : 0, stream);
    RT(rt_d2h(e->h_res, e->d_res_blk, res_words * 4, stream));
    return BO_OK;
}

static int result_unpack(bo_engine *e, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx, int32_t *best_move,
                         int32_t *total_visits) {
    const size_t G = (size_t)e->d.c.G;
    const int *h = e->h_res;
    if (res_n) memcpy(res_n, h, G * 4);
    if (best_idx) memcpy(best_idx, h + G, G * 4);
    if (best_move) memcpy(best_move, h + 2 * G, G * 4);
    if (total_visits) memcpy(total_visits, h + 3 * G, G * 4);
    if (res_idx) memcpy(res_idx, h + 4 * G, G * BO_RES_CAP * 4);
    if (res_val) memcpy(res_val, h + 4 * G + G * BO_RES_CAP, G * BO_RES_CAP * 4);
    e->watch_seen |= h[G * (4 + 2 * (size_t)BO_RES_CAP)];
    return BO_OK;
}

// A device status word of the evaluate stage (the split-precision tower's "an activation left the fp16 range", bo_nn_tower_word)
// rides along with every fetched result block: the ply's one host round trip checks it, no copy or wait of its own.
extern "C" int bo_engine_watch(bo_engine *e, int32_t *dev_word) { return bo_engine_watch_words(e, dev_word, 1); }
// n_words = 2: the two words of bo_nn_b1_word ([timeout code | saturation flag]); the copy is word 0 | (word 1 != 0) << 16
extern "C" int bo_engine_watch_words(bo_engine *e, int32_t *dev_words, int32_t n_words) {
    if (!e || n_words < 1 || n_words > 2) return fail(BO_E_ARG, "bad arguments");
    e->d.watch = dev_words;
    e->d.watch_n = n_words;
    return BO_OK;
}
// OR of the watched word over every result block fetched since the last call with clear != 0 (host state; nothing is enqueued)
extern "C" int bo_engine_watch_seen(bo_engine *e, int32_t *seen_out, int32_t clear) {
    if (!e || !seen_out) return fail(BO_E_ARG, "bad arguments");
    *seen_out = e->watch_seen;
    if (clear) e->watch_seen = 0;
    return BO_OK;
}

// The result block and the searches' state enqueued BEHIND the searches' last expected step, without waiting: when the caller later
// finds the stream idle, bo_selfplay_turn(flags 2 | 8) reads both from pinned memory and needs no round trip of its own (if a search
// turns out to need one more evaluation, the caller steps and prefetches again).  Any bo_step after it invalidates it.
extern "C" int bo_search_result_prefetch(bo_engine *e, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    int rc = result_enqueue(e, stream, true);
    if (rc) return rc;
    e->prefetch_valid = true;
    return BO_OK;
}

extern "C" int bo_search_result(bo_engine *e, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx,
                                int32_t *best_move, int32_t *total_visits, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    int rc = result_enqueue(e, stream);
    if (rc) return rc;
    RT(rt_sync(stream));
    return result_unpack(e, res_n, res_idx, res_val, best_idx, best_move, total_visits);
}

extern "C" int bo_play(bo_engine *e, const int32_t *action, void *stream) {
    if (!e || !action) return fail(BO_E_ARG, "null argument");
    if (e->ship) {  // the kernel reads the actions from pinned host memory itself (two buffers: the previous call's kernel may not have run yet)
        const int half = (e->action_flip ^= 1);
        int *slot = e->h_action + (size_t)half * e->d.c.G;
        if (e->ev_action_made[half]) RT(rt_event_sync(e->ev_action[half]));  // (the kernel that read this half two calls ago has run: normally long since)
        memcpy(slot, action, (size_t)e->d.c.G * 4);
        RT(RT_LAUNCH(bo_k_play, e->d.c.G, stream, e->d, (const int *)slot));
        if (!e->ev_action_made[half]) { RT(rt_event_create(&e->ev_action[half])); e->ev_action_made[half] = true; }
        RT(rt_event_record(e->ev_action[half], stream));
    } else {
        RT(rt_h2d(e->d_action, action, (size_t)e->d.c.G * 4, stream));
        RT(RT_LAUNCH(bo_k_play, e->d.c.G, stream, e->d, (const int *)e->d_action));
    }
    if (e->fast) RT(RT_LAUNCH(bo_k_fw_reroot, e->d.c.G, stream, e->d, e->f, e->fast_reuse));  // tree reuse: the played child becomes the root
    return BO_OK;
}

// ---- native per-move host work (bit-compatible with numpy.random.RandomState; bo_hostrng.h) ----------------------
// The per-game streams are independent, so the G Dirichlet draws / temperature samples of a ply are spread over a few
// persistent worker threads (the GPU idles during this phase: 160 us single-threaded for 256 roots).
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
namespace {
struct BoPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)> *job = nullptr;
    int n_items = 0, chunk = 1, generation = 0, pending = 0;
    std::atomic<int> next{0};
    bool stop = false;
    pid_t owner = getpid();
    explicit BoPool(int n) {
        for (int i = 0; i < n; i++) workers.emplace_back([this] { loop(); });
    }
    ~BoPool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv_go.notify_all();
        for (auto &t : workers) t.join();
    }
    void drain() {
        for (;;) {
            const int b = next.fetch_add(chunk);
            if (b >= n_items) return;
            const int e = b + chunk < n_items ? b + chunk : n_items;
            for (int i = b; i < e; i++) (*job)(i);
        }
    }
    void loop() {
        int seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m);
                cv_go.wait(l, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            drain();
            { std::lock_guard<std::mutex> l(m); pending--; }
            cv_done.notify_one();
        }
    }
    // light: microseconds of work in total (sampling one of <= 2 moves per game, flag bookkeeping) -- inline: waking the workers
    // and waiting for every one of them to report back costs a futex round trip (~30-50 us) the device idles through
    void run(int n, const std::function<void(int)> &f, bool light = false) {
        // (a fork()ed child inherits the object but not the threads: it works inline)
        // (fewer than 256 items -- a cohort's 64 roots take ~40 us inline -- are not worth waking the workers for: their wake-up is ~50 us
        //  at best and milliseconds when a worker's core sleeps deeply or is busy: p99 2.9 ms of bo_selfplay_noise, profiles/r05_logs/s16_plyprof.log)
        if (workers.empty() || n < 256 || (light && n <= 2048) || getpid() != owner) { for (int i = 0; i < n; i++) f(i); return; }
        {
            std::lock_guard<std::mutex> l(m);
            job = &f; n_items = n; chunk = 8; next = 0; pending = (int)workers.size(); generation++;
        }
        cv_go.notify_all();
        drain();  // the calling thread works too
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [&] { return pending == 0; });
    }
};
BoPool &host_pool() {
    // never destroyed: the workers end with the process (a destructor would have to join threads that a fork()ed child
    // does not have, or run during interpreter shutdown)
    static BoPool *pool = new BoPool([] {
        const char *v = getenv("BO_HOST_THREADS");
        int n = v ? atoi(v) : 4;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw / 2) n = hw / 2;  // leave room for the other ranks of a node
        return n < 1 ? 0 : n - 1;
    }());
    return *pool;
}
}  // namespace

extern "C" int bo_rng_seed(bo_engine *e, int slot, uint32_t seed) {
    if (!e || slot < 0 || slot >= e->d.c.G) return fail(BO_E_ARG, "bad slot");
    hr_seed(&e->rng[slot], seed);
    if ((int)e->noise_pending.size() == e->d.c.G) e->noise_pending[slot] = 0;  // a new game: the old root's deferred draw is void
    return BO_OK;
}

extern "C" int bo_rng_state(bo_engine *e, int slot, int set, uint32_t *key624, int32_t *pos, int32_t *has_gauss, double *gauss) {
    if (!e || slot < 0 || slot >= e->d.c.G || !key624 || !pos || !has_gauss || !gauss) return fail(BO_E_ARG, "bad arguments");
    HostRng &r = e->rng[slot];
    if (set) {
        memcpy(r.key, key624, sizeof(r.key));
        r.pos = *pos; r.has_gauss = *has_gauss; r.gauss = *gauss;
    } else {
        memcpy(key624, r.key, sizeof(r.key));
        *pos = r.pos; *has_gauss = r.has_gauss; *gauss = r.gauss;
    }
    return BO_OK;
}

// fetched: the result block is already in e->h_res (bo_selfplay_turn asked for it together with the searches' state)
static int selfplay_sample_impl(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold, double t_initial,
                                double t_final, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx,
                                int32_t *action_out, bool fetched, void *stream) {
    if (!e || !active || !move_number || !res_n || !res_idx || !res_val || !action_out) return fail(BO_E_ARG, "null argument");
    int rc = fetched ? result_unpack(e, res_n, res_idx, res_val, best_idx, nullptr, nullptr)
                     : bo_search_result(e, res_n, res_idx, res_val, best_idx, nullptr, nullptr, stream);
    if (rc) return rc;
    const int G = e->d.c.G;
    host_pool().run(G, [&](int g) {
        if (!active[g]) { action_out[g] = -1; return; }
        const int32_t *ri = res_idx + (size_t)g * BO_RES_CAP;
        const float *rv = res_val + (size_t)g * BO_RES_CAP;
        const int a = e->fast ? hr_select_action_general(&e->rng[g], res_n[g], ri, rv, move_number[g], threshold, t_initial, t_final)
                              : hr_select_action(&e->rng[g], res_n[g], ri, rv, move_number[g], threshold, t_initial, t_final);
        action_out[g] = a >= 0 ? a : -3;  // -3: not sparse enough, the caller samples with the dense NumPy mirror
    }, !e->fast);  // reference semantics: pi has <= 2 entries, a few dozen nanoseconds per game
    return BO_OK;
}

extern "C" int bo_selfplay_sample(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold,
                                  double t_initial, double t_final, int32_t *res_n, int32_t *res_idx, float *res_val,
                                  int32_t *best_idx, int32_t *action_out, void *stream) {
    return selfplay_sample_impl(e, active, move_number, threshold, t_initial, t_final, res_n, res_idx, res_val, best_idx, action_out, false, stream);
}

static int selfplay_begin_impl(bo_engine *e, const int32_t *want, float *nn_in_dev, int32_t *n_legal_out, int32_t *terminal_out,
                               int32_t *go_out, bool defer_noise, void *stream) {
    if (!e || !want || !nn_in_dev) return fail(BO_E_ARG, "null argument");
    const int G = e->d.c.G;
    int rc = bo_root_info(e, e->h_nl.data(), e->h_term.data(), nullptr, stream);
    if (rc) return rc;
    const double alpha = e->cfg.dirichlet_alpha;
    if ((int)e->noise_pending.size() != G) e->noise_pending.assign(G, 0);
    host_pool().run(G, [&](int g) {
        const int go = want[g] && e->h_term[g] == 0;
        e->h_go[g] = go;
        if (want[g]) e->noise_pending[g] = 0;
        if (go && alpha > 0) {
            if (defer_noise) e->noise_pending[g] = 1;
            else hr_dirichlet(&e->rng[g], alpha, e->h_nl[g], &e->h_noise[(size_t)g * BO_MAX_MOVES]);  // mcts.py:192
        }
        if (n_legal_out) n_legal_out[g] = e->h_nl[g];
        if (terminal_out) terminal_out[g] = e->h_term[g];
        if (go_out) go_out[g] = go;
    }, defer_noise || alpha <= 0);  // (without the Dirichlet draws this is flag bookkeeping)
    e->nl_valid = true;
    rc = search_begin_impl(e, e->h_go, (alpha > 0 && !defer_noise) ? e->h_noise : nullptr, defer_noise, nn_in_dev, stream);
    if (rc) return rc;
    return bo_step(e, nullptr, nullptr, BO_POLICY_NONE, nn_in_dev, stream);
}

extern "C" int bo_selfplay_begin(bo_engine *e, const int32_t *want, float *nn_in_dev, int32_t *n_legal_out, int32_t *terminal_out,
                                 int32_t *go_out, void *stream) {
    if (e && e->begin_lazy) return fail(BO_E_STATE, "bo_selfplay_begun has not collected the previous turn's roots yet");
    return selfplay_begin_impl(e, want, nn_in_dev, n_legal_out, terminal_out, go_out, false, stream);
}

// The begin of bo_selfplay_turn(flag 4): nothing here waits for the device.  go is decided by the kernel (wanted and not
// terminal); the roots' state travels behind it and is collected by bo_selfplay_begun once the caller has enqueued the first
// evaluation -- the device goes from the played moves straight into that evaluation instead of idling through a host round trip.
static int selfplay_begin_lazy(bo_engine *e, const int32_t *want, float *nn_in_dev, void *stream) {
    if (e->fast) return fail(BO_E_CONFIG, "bo_selfplay_turn: flag 4 needs a reference-semantics engine");
    const size_t G = (size_t)e->d.c.G;
    e->lazy_want.assign(want, want + G);
    memcpy(e->h_go, want, G * 4);
    if (e->ship) { int rcs = ship(e, e->d_go, e->h_go, G, nullptr, nullptr, 0, stream); if (rcs) return rcs; }  // (h_go is next written by bo_selfplay_begun, behind ev_begin)
    else RT(rt_h2d(e->d_go, e->h_go, G * 4, stream));
    e->nl_valid = false;
    RT(RT_LAUNCH(bo_k_search_begin_want, e->d.c.G, stream, e->d, (const int *)e->d_go, nn_in_dev));
    int rc = bo_step(e, nullptr, nullptr, BO_POLICY_NONE, nn_in_dev, stream);
    if (rc) return rc;
    if (e->ship) { rc = ship(e, e->h_info + 2 * G, (const int *)e->d_info_blk + 2 * G, 3 * G, nullptr, nullptr, 0, stream); if (rc) return rc; }
    else RT(rt_d2h(e->h_info + 2 * G, e->d_info_blk + 2 * G, 3 * G * 4, stream));  // [root_nlegal | root_term | ply]
    if (!e->ev_begin_made) { RT(rt_event_create(&e->ev_begin)); e->ev_begin_made = true; }
    RT(rt_event_record(e->ev_begin, stream));
    e->begin_lazy = true;
    return BO_OK;
}

// Collect what bo_selfplay_turn(flag 4) left on its way: the new roots' legal-move counts, terminal codes and go flags (as
// bo_selfplay_begin reports them), and which roots still need their Dirichlet draw (bo_selfplay_noise).  Waits for the copy
// behind the begin kernels only, not for work enqueued after the turn.
extern "C" int bo_selfplay_begun(bo_engine *e, int32_t *n_legal_out, int32_t *terminal_out, int32_t *go_out) {
    if (!e) return fail(BO_E_ARG, "null engine");
    if (!e->begin_lazy) return fail(BO_E_STATE, "no bo_selfplay_turn(flag 4) outstanding");
    RT(rt_event_sync(e->ev_begin));
    const int G = e->d.c.G;
    const double alpha = e->cfg.dirichlet_alpha;
    if ((int)e->noise_pending.size() != G) e->noise_pending.assign(G, 0);
    for (int g = 0; g < G; g++) {
        e->h_nl[g] = e->h_info[2 * G + g];
        e->h_term[g] = e->h_info[3 * G + g];
        const int go = e->lazy_want[g] && e->h_term[g] == 0;
        e->h_go[g] = go;
        if (e->lazy_want[g]) e->noise_pending[g] = 0;
        if (go && alpha > 0) e->noise_pending[g] = 1;
        if (n_legal_out) n_legal_out[g] = e->h_nl[g];
        if (terminal_out) terminal_out[g] = e->h_term[g];
        if (go_out) go_out[g] = go;
    }
    e->nl_valid = true;
    e->begin_lazy = false;
    return BO_OK;
}

// The Dirichlet draws bo_selfplay_turn(defer_noise = 1) left out, and their upload: call it after the root evaluation's
// network forward has been enqueued (the host work then overlaps it) and before the bo_step that consumes that evaluation.
extern "C" int bo_selfplay_noise(bo_engine *e, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    if (e->begin_lazy) return fail(BO_E_STATE, "bo_selfplay_noise: call bo_selfplay_begun first");
    const int G = e->d.c.G;
    const double alpha = e->cfg.dirichlet_alpha;
    if ((int)e->noise_pending.size() != G || alpha <= 0) return BO_OK;
    bool any = false;
    for (int g = 0; g < G; g++) any = any || e->noise_pending[g];
    if (!any) return BO_OK;
    host_pool().run(G, [&](int g) {
        if (e->noise_pending[g]) hr_dirichlet(&e->rng[g], alpha, e->h_nl[g], &e->h_noise[(size_t)g * BO_MAX_MOVES]);
    });
    e->nl_valid = true;
    int rc = upload_noise(e, e->noise_pending.data(), stream);
    e->nl_valid = false;
    std::fill(e->noise_pending.begin(), e->noise_pending.end(), 0);
    return rc;
}

// One host round trip per ply: sample the moves of the finished searches, play them, and begin the next searches
// (self_play.py:121-184 followed by the next iteration's mcts.py:185-203).  Per game the RNG stream order is unchanged:
// the temperature sample, then the Dirichlet draw of the new root.
extern "C" int bo_selfplay_turn(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold, double t_initial,
                                double t_final, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx, int32_t *action_out,
                                const int32_t *want_next, float *nn_in_dev, int32_t *n_legal_out, int32_t *terminal_out, int32_t *go_out,
                                int32_t defer_noise, int32_t *completed, void *stream) {
    if (!e || !want_next || !nn_in_dev || !completed) return fail(BO_E_ARG, "null argument");
    *completed = 0;
    const bool poll_first = (defer_noise & 2) != 0, lazy_begin = (defer_noise & 4) != 0, prefetched = (defer_noise & 8) != 0;
    defer_noise &= 1;
    if (prefetched && !poll_first) return fail(BO_E_ARG, "bo_selfplay_turn: flag 8 (result block prefetched) goes with flag 2");
    if (lazy_begin && !defer_noise) return fail(BO_E_ARG, "bo_selfplay_turn: flag 4 (begin without waiting) needs flag 1 (noise later)");
    if (e->begin_lazy) return fail(BO_E_STATE, "bo_selfplay_begun has not collected the previous turn's roots yet");
    if (e->turn_outstanding) return fail(BO_E_STATE, "bo_selfplay_autoturn_collect has not collected the device's turn yet");
    int rc;
    if (poll_first) {
        // "are all searches finished?" and their results in ONE round trip: the result kernel runs behind the last expected
        // step without waiting for the answer (it only reads the trees; if a search needs one more evaluation -- rare -- its
        // block is fetched again by the next call)
        const size_t G = (size_t)e->d.c.G;
        if (!(prefetched && e->prefetch_valid)) {
            rc = result_enqueue(e, stream, true);  // (+ [phase | req_node])
            if (rc) return rc;
        }
        e->prefetch_valid = false;
        RT(rt_sync(stream));
        for (size_t g = 0; g < G; g++)
            if (e->h_info[g] == PH_RUN) { *completed = -1; return BO_OK; }
    }
    rc = selfplay_sample_impl(e, active, move_number, threshold, t_initial, t_final, res_n, res_idx, res_val, best_idx, action_out, poll_first, stream);
    if (rc) return rc;
    for (int g = 0; g < e->d.c.G; g++)
        if (action_out[g] == -3) return BO_OK;  // a pi too dense for the native sampler: the caller samples it, then plays and begins
    rc = bo_play(e, action_out, stream);
    if (rc) return rc;
    rc = lazy_begin ? selfplay_begin_lazy(e, want_next, nn_in_dev, stream)
                    : selfplay_begin_impl(e, want_next, nn_in_dev, n_legal_out, terminal_out, go_out, defer_noise != 0, stream);
    if (rc) return rc;
    *completed = lazy_begin ? 2 : 1;
    return BO_OK;
}

// ---- the turn on the device (ABI 5) ------------------------------------------------------------------------------------------
// bo_selfplay_turn's work without the host in the device's way: the random draw of every move is made AHEAD (per game the stream order
// stays Dirichlet of this search -> choice of this move -> Dirichlet of the next search), the kernels bo_k_turn_sample / bo_k_turn_play are
// enqueued behind the searches' last expected step, and their outputs travel to pinned memory behind them.  Nothing here waits.
extern "C" int bo_selfplay_autoturn(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold, double t_initial,
                                    double t_final, const int32_t *want_next, float *nn_in_dev, int32_t redo, void *stream) {
    if (!e || !active || !move_number || !want_next || !nn_in_dev) return fail(BO_E_ARG, "null argument");
    if (e->fast || e->d.c.root_m != 1 || e->d.c.S < 1)
        return fail(BO_E_CONFIG, "bo_selfplay_autoturn: reference-semantics engines whose root keeps <= 2 children (int(WIDEN_COEFF) == 1, NUM_SIMULATIONS >= 1)");
    if (!(fabs(t_initial - 1.0) < 1e-6) || t_final == 0.0 || !(t_final > 0.0))
        return fail(BO_E_CONFIG, "bo_selfplay_autoturn: TEMPERATURE_INITIAL must be 1 and TEMPERATURE_FINAL > 0 (other settings: bo_selfplay_turn)");
    if (!e->ship) return fail(BO_E_CONFIG, "bo_selfplay_autoturn is not available with BETAONE_TURN_COPIES=1");
    if (e->begin_lazy) return fail(BO_E_STATE, "bo_selfplay_begun has not collected the previous turn's roots yet");
    if (e->turn_outstanding) return fail(BO_E_STATE, "bo_selfplay_autoturn: the previous turn has not been collected");
    const int G = e->d.c.G, S = e->d.c.S;
    if (e->turn_pw_tfinal != t_final) {  // apply_temperature's np.power (self_play.py:37) for every possible pi entry f32(c / S): the host's libm, once
        e->turn_pw_host.resize((size_t)S + 1);
        for (int c = 0; c <= S; c++) {
            const float p = (float)((double)c / (double)S);  // mcts.py:273
            double x = pow((double)p, 1.0 / t_final);
            if (!isfinite(x)) x = 0.0;
            e->turn_pw_host[c] = x;
        }
        RT(rt_h2d(e->d_turn_pw, e->turn_pw_host.data(), ((size_t)S + 1) * sizeof(double), stream));
        RT(rt_sync(stream));  // (once per temperature: the staging vector may be rebuilt by the next change)
        e->turn_pw_tfinal = t_final;
    }
    if (!redo) {
        for (int g = 0; g < G; g++) {
            const double temp = move_number[g] < threshold ? t_initial : t_final;
            int fl = (active[g] ? 1 : 0) | (want_next[g] ? 4 : 0);
            if (active[g] && !(fabs(temp - 1.0) < 1e-6)) fl |= 2;
            e->h_turn_flags[g] = fl;
            if (active[g]) e->h_turn_u[g] = hr_double(&e->rng[g]);  // RandomState.choice's random_sample() (self_play.py:73)
        }
        e->turn_want.assign(want_next, want_next + G);
    }
    TurnArgs a;
    a.u = e->h_turn_u; a.flags = e->h_turn_flags; a.pw = e->d_turn_pw;
    a.action = e->d_turn_action; a.state = e->d_turn_state; a.cres = e->d_turn_cres;
    e->prefetch_valid = false;
    e->nl_valid = false;
    RT(RT_LAUNCH(bo_k_turn_sample, G, stream, e->d, a));
    RT(RT_LAUNCH(bo_k_turn_play, G, stream, e->d, a, nn_in_dev));
    const size_t Gs = (size_t)G, watch_off = Gs * (4 + 2 * (size_t)BO_RES_CAP);
    int rc = ship(e, e->h_info + 2 * Gs, (const int *)e->d_info_blk + 2 * Gs, 13 * Gs, e->h_res + watch_off, (const int *)e->d_res_blk + watch_off, 4, stream);
    if (rc) return rc;
    if (!e->ev_turn_made) { RT(rt_event_create(&e->ev_turn)); e->ev_turn_made = true; }
    RT(rt_event_record(e->ev_turn, stream));
    rc = bo_step(e, nullptr, nullptr, BO_POLICY_NONE, nn_in_dev, stream);  // the begun searches' first step (a no-op for every game if the turn did not happen)
    if (rc) return rc;
    e->turn_outstanding = true;
    return BO_OK;
}

// 1: the event behind the turn's outputs has been reached (bo_selfplay_autoturn_collect will not wait), 0: not yet
extern "C" int bo_selfplay_autoturn_ready(bo_engine *e, int32_t *ready_out) {
    if (!e || !ready_out) return fail(BO_E_ARG, "null argument");
    if (!e->turn_outstanding) return fail(BO_E_STATE, "no bo_selfplay_autoturn outstanding");
#if defined(BO_WAVE_EMU)
    *ready_out = 1;
#else
    const hipError_t q = hipEventQuery(e->ev_turn);
    if (q != hipSuccess && q != hipErrorNotReady) return fail(BO_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
    *ready_out = q == hipSuccess ? 1 : 0;
#endif
    return BO_OK;
}

extern "C" int bo_selfplay_autoturn_collect(bo_engine *e, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx, int32_t *action_out,
                                            int32_t *n_legal_out, int32_t *terminal_out, int32_t *go_out, int32_t *completed) {
    if (!e || !res_n || !res_idx || !res_val || !action_out || !completed) return fail(BO_E_ARG, "null argument");
    if (!e->turn_outstanding) return fail(BO_E_STATE, "no bo_selfplay_autoturn outstanding");
    RT(rt_event_sync(e->ev_turn));
    e->turn_outstanding = false;
    *completed = 0;
    const int G = e->d.c.G;
    const size_t Gs = (size_t)G;
    const int *act = e->h_info + 5 * Gs, *st = e->h_info + 6 * Gs, *cres = e->h_info + 7 * Gs;
    e->watch_seen |= e->h_res[Gs * (4 + 2 * (size_t)BO_RES_CAP)];
    bool running = false;
    for (int g = 0; g < G; g++) {
        if (st[g] == 2) return fail(BO_E_STATE, "bo_selfplay_autoturn: a search result the device sampler does not cover (slot " + std::to_string(g) + ")");
        running = running || st[g] == 1 || st[g] == 3;  // (3: the watched fault word was set -- bo_engine_watch_seen says so; nothing was played)
    }
    if (running) { *completed = -1; return BO_OK; }  // nothing was played: bo_step once more, then bo_selfplay_autoturn(redo = 1)
    const double alpha = e->cfg.dirichlet_alpha;
    if ((int)e->noise_pending.size() != G) e->noise_pending.assign(G, 0);
    for (int g = 0; g < G; g++) {
        const int *c = cres + (size_t)g * 8;
        action_out[g] = act[g];
        if (e->h_turn_flags[g] & 1) {
            res_n[g] = c[0];
            if (best_idx) best_idx[g] = c[1];
            res_idx[(size_t)g * BO_RES_CAP] = c[4]; res_idx[(size_t)g * BO_RES_CAP + 1] = c[5];
            memcpy(&res_val[(size_t)g * BO_RES_CAP], &c[6], 4); memcpy(&res_val[(size_t)g * BO_RES_CAP + 1], &c[7], 4);
        } else res_n[g] = 0;
        e->h_nl[g] = e->h_info[2 * Gs + g];
        e->h_term[g] = e->h_info[3 * Gs + g];
        const int go = e->turn_want[g] && e->h_term[g] == 0;
        e->h_go[g] = go;
        if (e->turn_want[g]) e->noise_pending[g] = 0;
        if (go && alpha > 0) e->noise_pending[g] = 1;
        if (n_legal_out) n_legal_out[g] = e->h_nl[g];
        if (terminal_out) terminal_out[g] = e->h_term[g];
        if (go_out) go_out[g] = go;
    }
    e->nl_valid = true;
    *completed = 1;
    return BO_OK;
}

// ---- records ------------------------------------------------------------------------------------------------
extern "C" int bo_game_export(bo_engine *e, int slot, bo_position *positions, int32_t *moves, int32_t cap, int32_t *n_plies,
                              void *stream) {
    if (!e || slot < 0 || slot >= e->d.c.G || !n_plies) return fail(BO_E_ARG, "bad arguments");
    const EngCfg &c = e->d.c;
    // ply count and (up to cap) positions / moves in one round trip: the caller's cap is normally exact (it counted the plies)
    const int want = cap < c.PLY_CAP ? (cap > 0 ? cap : 0) : c.PLY_CAP;
    int ply = 0;
    std::vector<DPos> gp((size_t)want);
    std::vector<bo_mv> mv((size_t)want);
    RT(rt_d2h(&ply, e->d.ply + slot, 4, stream));
    if (want > 0 && (positions || moves)) {
        RT(rt_d2h(gp.data(), e->d.gpos + (size_t)slot * c.PLY_CAP, gp.size() * sizeof(DPos), stream));
        RT(rt_d2h(mv.data(), e->d.played + (size_t)slot * c.PLY_CAP, mv.size() * sizeof(bo_mv), stream));
    }
    RT(rt_sync(stream));
    *n_plies = ply;
    if (!positions && !moves) return BO_OK;
    if (ply + 1 > cap) return fail(BO_E_ARG, "export buffer too small");
    for (int i = 0; i <= ply; i++) {
        if (positions) to_abi(gp[i], &positions[i]);
        if (moves && i < ply) moves[i] = mv[i];
    }
    return BO_OK;
}

extern "C" int bo_game_encode(bo_engine *e, int slot, int first, int n, float *out_dev, void *stream) {
    if (!e || slot < 0 || slot >= e->d.c.G || n < 1 || first < 0 || !out_dev) return fail(BO_E_ARG, "bad arguments");
    RT(RT_LAUNCH(bo_k_encode_game, n, stream, e->d, slot, first, out_dev));
    return BO_OK;
}

extern "C" int bo_records_encode(int n_positions, const bo_position *positions, int first, int n, float *out_dev, void *stream) {
    if (n_positions < 1 || !positions || first < 0 || n < 1 || first + n > n_positions || !out_dev)
        return fail(BO_E_ARG, "bad arguments");
    std::vector<DPos> hp((size_t)n_positions);
    for (int i = 0; i < n_positions; i++) hp[i] = from_abi(positions[i]);
    void *dp = nullptr;
    int rc = rt_malloc(&dp, hp.size() * sizeof(DPos));
    if (!rc) rc = rt_h2d(dp, hp.data(), hp.size() * sizeof(DPos), stream);
    if (!rc) rc = RT_LAUNCH(bo_k_encode_positions, n, stream, (const DPos *)dp, n_positions, first, out_dev);
    int rc2 = rt_sync(stream);
    rt_free(dp);
    if (rc || rc2) return fail(BO_E_HIP, std::string("bo_records_encode: ") + rt_errstr(rc ? rc : rc2));
    return BO_OK;
}

// ---- GPU-resident replay buffer (bo_replay.h) ------------------------------------------------------
// Storage is a ring of position slots; a game of n records takes n + 1 consecutive slots (its positions, the final one included: the
// END-of-game tracker counts it), so a sampled ply finds its history blocks next to it.  Games are evicted oldest first when the ring
// comes round.  Host side: the table of resident games; device side: SoA arrays indexed by slot.
struct bo_replay_s {
    int device = 0, W = 2;
    int64_t cap = 0, head = 0, n_records = 0;
    DPos *pos = nullptr;
    int *rep = nullptr, *pi_n = nullptr, *pi_idx = nullptr, *s_slot = nullptr, *s_k = nullptr;
    float *pi_val = nullptr, *z = nullptr;
    int s_cap = 0;
    struct Game { int64_t start; int32_t n_rec; int32_t game_id; };
    std::deque<Game> games;
    std::vector<int64_t> prefix;   // prefix[i] = records of games[0..i) (rebuilt after an add)
    std::vector<int> h_slot, h_k;
};

extern "C" int bo_replay_create(int64_t capacity_positions, int pi_width, int device, bo_replay **out) {
    if (!out || capacity_positions < 2 || capacity_positions > (int64_t)0x7fffffff || pi_width < 1 || pi_width > BO_RES_CAP) return fail(BO_E_ARG, "bo_replay_create: bad arguments");
    RT(rt_set_device(device));
    bo_replay *r = new bo_replay();
    r->device = device; r->W = pi_width; r->cap = capacity_positions;
    const size_t n = (size_t)capacity_positions;
    int rc = rt_malloc((void **)&r->pos, n * sizeof(DPos));
    if (!rc) rc = rt_malloc((void **)&r->rep, n * 4);
    if (!rc) rc = rt_malloc((void **)&r->pi_n, n * 4);
    if (!rc) rc = rt_malloc((void **)&r->pi_idx, n * (size_t)pi_width * 4);
    if (!rc) rc = rt_malloc((void **)&r->pi_val, n * (size_t)pi_width * 4);
    if (!rc) rc = rt_malloc((void **)&r->z, n * 4);
    if (rc) {
        rt_free(r->pos); rt_free(r->rep); rt_free(r->pi_n); rt_free(r->pi_idx); rt_free(r->pi_val); rt_free(r->z);
        delete r;
        return fail(BO_E_HIP, std::string("bo_replay_create: ") + rt_errstr(rc));
    }
    *out = r;
    return BO_OK;
}

extern "C" void bo_replay_destroy(bo_replay *r) {
    if (!r) return;
    rt_free(r->pos); rt_free(r->rep); rt_free(r->pi_n); rt_free(r->pi_idx); rt_free(r->pi_val); rt_free(r->z); rt_free(r->s_slot); rt_free(r->s_k);
    delete r;
}

// One finished game: positions[0 .. n_records] (the record wire format's list: position i is the one before move i, the last is the
// final position), pi of record i = entries pi_ptr[i] .. pi_ptr[i + 1] of (pi_idx, pi_val), z[i] as the reference stores it
// (self_play.py:202: the outcome from the point of view of the side to move, sign of zero included).  Evicts the oldest games that
// are in the way; *evicted_records (may be NULL) = how many records that cost.  Synchronises `stream` (host staging is freed).
extern "C" int bo_replay_add_game(bo_replay *r, int32_t game_id, const bo_position *positions, int32_t n_records, const int32_t *pi_ptr,
                                  const int32_t *pi_idx, const float *pi_val, const float *z, int64_t *evicted_records, void *stream) {
    if (!r || !positions || n_records < 0 || (n_records && (!pi_ptr || !pi_idx || !pi_val || !z))) return fail(BO_E_ARG, "bo_replay_add_game: bad arguments");
    if (evicted_records) *evicted_records = 0;
    if (n_records == 0) return BO_OK;  // (a start position that was already over: no examples, self_play.py:101)
    const int64_t need = (int64_t)n_records + 1;
    if (need > r->cap) return fail(BO_E_ARG, "bo_replay_add_game: the game is longer than the buffer");
    for (int i = 0; i < n_records; i++)
        if (pi_ptr[i + 1] - pi_ptr[i] > r->W || pi_ptr[i + 1] < pi_ptr[i]) return fail(BO_E_ARG, "bo_replay_add_game: a pi has more entries than the buffer's pi_width");
    int64_t lost = 0;
    auto evict_front = [&]() { lost += r->games.front().n_rec; r->n_records -= r->games.front().n_rec; r->games.pop_front(); };
    if (r->head + need > r->cap) {  // does not fit behind the newest game: the games still living in that tail go, the ring comes round
        while (!r->games.empty() && r->games.front().start >= r->head) evict_front();
        r->head = 0;
    }
    while (!r->games.empty() && r->games.front().start >= r->head && r->games.front().start < r->head + need) evict_front();
    const size_t n = (size_t)need;
    std::vector<DPos> hp(n);
    for (size_t i = 0; i < n; i++) hp[i] = from_abi(positions[i]);
    std::vector<int> hn(n, 0), hi(n * (size_t)r->W, 0);
    std::vector<float> hv(n * (size_t)r->W, 0.0f), hz(n, 0.0f);
    for (int i = 0; i < n_records; i++) {
        hn[i] = pi_ptr[i + 1] - pi_ptr[i];
        for (int e = 0; e < hn[i]; e++) { hi[(size_t)i * r->W + e] = pi_idx[pi_ptr[i] + e]; hv[(size_t)i * r->W + e] = pi_val[pi_ptr[i] + e]; }
        hz[i] = z[i];
    }
    const size_t o = (size_t)r->head;
    int rc = rt_h2d(r->pos + o, hp.data(), n * sizeof(DPos), stream);
    if (!rc) rc = rt_h2d(r->pi_n + o, hn.data(), n * 4, stream);
    if (!rc) rc = rt_h2d(r->pi_idx + o * r->W, hi.data(), n * (size_t)r->W * 4, stream);
    if (!rc) rc = rt_h2d(r->pi_val + o * r->W, hv.data(), n * (size_t)r->W * 4, stream);
    if (!rc) rc = rt_h2d(r->z + o, hz.data(), n * 4, stream);
    if (!rc) rc = RT_LAUNCH(bo_k_replay_counts, (int)need, stream, (const DPos *)(r->pos + o), (int)need, r->rep + o);
    const int rc2 = rt_sync(stream);
    if (rc || rc2) return fail(BO_E_HIP, std::string("bo_replay_add_game: ") + rt_errstr(rc ? rc : rc2));
    r->games.push_back({r->head, n_records, game_id});
    r->head += need;
    r->n_records += n_records;
    r->prefix.clear();
    if (evicted_records) *evicted_records = lost;
    return BO_OK;
}

extern "C" int bo_replay_size(bo_replay *r, int64_t *n_records, int64_t *n_games) {
    if (!r) return fail(BO_E_ARG, "null handle");
    if (n_records) *n_records = r->n_records;
    if (n_games) *n_games = (int64_t)r->games.size();
    return BO_OK;
}

// A batch: record_index[i] in [0, records) counts the resident records oldest game first (ChessDataset's index space over
// load_recent_data's concatenation, train.py:179-219); states [n,120,8,8], pi [n,4672], z [n] are written on `stream` (no wait).
extern "C" int bo_replay_sample(bo_replay *r, int32_t n, const int64_t *record_index, float *states_dev, float *pi_dev, float *z_dev, void *stream) {
    if (!r || n < 1 || !record_index || !states_dev || !pi_dev || !z_dev) return fail(BO_E_ARG, "bo_replay_sample: bad arguments");
    if (r->prefix.empty()) {
        r->prefix.reserve(r->games.size() + 1);
        int64_t acc = 0;
        for (const auto &g : r->games) { r->prefix.push_back(acc); acc += g.n_rec; }
        r->prefix.push_back(acc);
    }
    r->h_slot.resize((size_t)n); r->h_k.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const int64_t q = record_index[i];
        if (q < 0 || q >= r->n_records) return fail(BO_E_ARG, "bo_replay_sample: record index out of range");
        const size_t g = (size_t)(std::upper_bound(r->prefix.begin(), r->prefix.end(), q) - r->prefix.begin()) - 1;
        const int k = (int)(q - r->prefix[g]);
        r->h_slot[(size_t)i] = (int)(r->games[g].start + k);
        r->h_k[(size_t)i] = k;
    }
    if (n > r->s_cap) {
        RT(rt_sync(stream));  // (the previous batch's index arrays may still be read)
        rt_free(r->s_slot); rt_free(r->s_k);
        r->s_slot = r->s_k = nullptr; r->s_cap = 0;
        RT(rt_malloc((void **)&r->s_slot, (size_t)n * 4));
        RT(rt_malloc((void **)&r->s_k, (size_t)n * 4));
        r->s_cap = n;
    }
    RT(rt_h2d(r->s_slot, r->h_slot.data(), (size_t)n * 4, stream));
    RT(rt_h2d(r->s_k, r->h_k.data(), (size_t)n * 4, stream));
    RT(RT_LAUNCH(bo_k_replay_encode, n, stream, (const DPos *)r->pos, (const int *)r->rep, (const int *)r->pi_n, (const int *)r->pi_idx,
                 (const float *)r->pi_val, (const float *)r->z, r->W, (const int *)r->s_slot, (const int *)r->s_k, states_dev, pi_dev, z_dev));
    // the pageable host index arrays are re-used by the next call: wait for their upload (the encode kernel itself is not waited for)
    RT(rt_sync(stream));
    return BO_OK;
}

// ---- introspection -----------------------------------------------------------------------------------------
extern "C" int bo_debug_tree(bo_engine *e, int slot, bo_node *out, int32_t cap, int32_t *n_nodes, void *stream) {
    if (!e || slot < 0 || slot >= e->d.c.G || !n_nodes) return fail(BO_E_ARG, "bad arguments");
    const EngCfg &c = e->d.c;
    int nn = 0;
    RT(rt_d2h(&nn, e->d.n_nodes + slot, 4, stream));
    RT(rt_sync(stream));
    if (e->fast) {  // child-block arena (bo_fastw.h): nodes in breadth-first order, children in record order
        const FastW &f = e->f;
        int head[FWC_HEAD];
        RT(rt_d2h(head, f.ctl + (size_t)slot * f.CS, sizeof(head), stream));
        RT(rt_sync(stream));
        const int cur = head[FWC_CUR], top = head[FWC_TOP];
        std::vector<WRec> A((size_t)top * BO_FW_GR);
        std::vector<bo_mv> M((size_t)top * BO_FW_GR);
        const size_t aoff = ((size_t)slot * 2 + (size_t)cur) * (size_t)f.NG * BO_FW_GR;
        RT(rt_d2h(A.data(), f.arena + aoff, A.size() * sizeof(WRec), stream));
        RT(rt_d2h(M.data(), f.amove + aoff, M.size() * sizeof(bo_mv), stream));
        RT(rt_sync(stream));
        std::vector<bo_node> nodes;
        std::vector<int> rec;  // record id of each node
        auto push = [&](int r, int parent) {
            bo_node nd;
            nd.parent = parent; nd.n_visits = A[r].n; nd.first_child = 0; nd.n_children = 0; nd.q_value = A[r].w; nd.prior = A[r].prior;
            nd.move = r == 0 ? 0 : M[r];
            nd.terminal = A[r].link == FW_MATE ? 1 : A[r].link == FW_DRAW ? 2 : A[r].link >= 0 ? 0 : -1;
            nodes.push_back(nd); rec.push_back(r);
        };
        push(0, -1);
        for (size_t i = 0; i < nodes.size(); i++) {
            const int link = A[rec[i]].link;
            if (link < 0) continue;
            const int first = link & BO_FW_LINK_MASK, ng = ((link >> 24) & 127) + 1;
            if (first < 1 + BO_FW_HG || (size_t)(first + ng) * BO_FW_GR > A.size()) return fail(BO_E_HIP, "corrupt child-run link");
            nodes[i].first_child = (int)nodes.size();
            for (int k = 0; k < ng * BO_FW_GR; k++)
                if (A[(size_t)first * BO_FW_GR + k].n >= 0) { push(first * BO_FW_GR + k, (int)i); nodes[i].n_children++; }
        }
        *n_nodes = (int)nodes.size();
        if (!out) return BO_OK;
        if ((int)nodes.size() > cap) return fail(BO_E_ARG, "tree buffer too small");
        memcpy(out, nodes.data(), nodes.size() * sizeof(bo_node));
        return BO_OK;
    }
    *n_nodes = nn;
    if (!out) return BO_OK;
    if (nn > cap) return fail(BO_E_ARG, "tree buffer too small");
    const size_t off = (size_t)slot * c.NCAP, N = (size_t)nn;
    std::vector<int> nv(N), pa(N), fc(N), nc(N);
    std::vector<float> q(N), pr(N);
    std::vector<bo_mv> mv(N);
    std::vector<signed char> tm(N);
    RT(rt_d2h(nv.data(), e->d.n_visits + off, N * 4, stream)); RT(rt_d2h(pa.data(), e->d.parent + off, N * 4, stream));
    RT(rt_d2h(fc.data(), e->d.first_child + off, N * 4, stream)); RT(rt_d2h(nc.data(), e->d.n_children + off, N * 4, stream));
    RT(rt_d2h(q.data(), e->d.q + off, N * 4, stream)); RT(rt_d2h(pr.data(), e->d.prior + off, N * 4, stream));
    RT(rt_d2h(mv.data(), e->d.move + off, N * sizeof(bo_mv), stream)); RT(rt_d2h(tm.data(), e->d.term + off, N, stream));
    RT(rt_sync(stream));
    for (size_t i = 0; i < N; i++) {
        out[i].parent = pa[i]; out[i].n_visits = nv[i]; out[i].first_child = fc[i]; out[i].n_children = nc[i];
        out[i].q_value = q[i]; out[i].prior = pr[i]; out[i].move = mv[i]; out[i].terminal = tm[i];
    }
    return BO_OK;
}

extern "C" int bo_debug_fast(bo_engine *e, int slot, int32_t *ctl_out, int32_t ctl_cap, int32_t *paths_out, void *stream) {
    if (!e || slot < 0 || slot >= e->d.c.G) return fail(BO_E_ARG, "bad arguments");
    if (!e->fast) return fail(BO_E_CONFIG, "bo_debug_fast: fast-mode engines only");
    const FastW &f = e->f;
    if (ctl_out) {
        if (ctl_cap < f.CS) return fail(BO_E_ARG, "control block buffer too small");
        RT(rt_d2h(ctl_out, f.ctl + (size_t)slot * f.CS, (size_t)f.CS * 4, stream));
    }
    std::vector<int32_t> dm;  // the device keeps the path rows depth-major (FW_PIDX); this call hands them out simulation-major [L][PATH_CAP] as it always has
    if (paths_out) {
        dm.resize((size_t)f.L * BO_FW_PATH_CAP);
        RT(rt_d2h(dm.data(), f.sim_path + (size_t)slot * f.L * BO_FW_PATH_CAP, dm.size() * 4, stream));
    }
    RT(rt_sync(stream));
    if (paths_out)
        for (int s = 0; s < f.L; s++)
            for (int d = 0; d < BO_FW_PATH_CAP; d++) paths_out[(size_t)s * BO_FW_PATH_CAP + d] = dm[FW_PIDX(f.L, s, d)];
    return BO_OK;
}

extern "C" int bo_fast_options(bo_engine *e, int32_t tree_reuse, int32_t games_per_halfwave, int32_t select_flags) {
    if (!e) return fail(BO_E_ARG, "null engine");
    if (!e->fast) return fail(BO_E_CONFIG, "bo_fast_options: fast-mode engines only");
    if (games_per_halfwave >= 0 && games_per_halfwave != 1 && games_per_halfwave != 2 && games_per_halfwave != 4)
        return fail(BO_E_ARG, "bo_fast_options: games_per_halfwave must be 1, 2 or 4");
    if (tree_reuse >= 0) e->fast_reuse = tree_reuse ? 1 : 0;
    if (games_per_halfwave >= 0) e->f.sel_ut = games_per_halfwave == 1 ? 2 : games_per_halfwave;  // (one game per half-wave is the form for more than 16 leaves per step)
    if (select_flags >= 0) e->f.sel_flags = select_flags & (FW_SEL_NT | FW_SEL_ROOT_IN_REGS | FW_SEL_DENSE | FW_SEL_LANE | FW_SEL_OCT | FW_SEL_QUAD);
    return BO_OK;
}

#if !defined(BO_WAVE_EMU)
__global__ void bo_k_nothing() {}
#endif
extern "C" int bo_event_pair_overhead(double *ms_out, int32_t samples, void *stream) {
    if (!ms_out || samples < 1 || samples > 256) return fail(BO_E_ARG, "bad arguments");
#if defined(BO_WAVE_EMU)
    (void)stream;
    *ms_out = 0.0;
    return BO_OK;
#else
    hipEvent_t e0, e1;
    RT((int)hipEventCreate(&e0));
    RT((int)hipEventCreate(&e1));
    // pair around ONE empty kernel = overhead + one empty kernel; around TWO = overhead + two: overhead = 2 * p1 - p2
    std::vector<float> v[2];
    int rc = 0;
    for (int i = 0; i < samples + 4 && !rc; i++) {  // (the first few launches load the code object and warm the queue)
        for (int k = 0; k < 2 && !rc; k++) {
            rc = (int)hipEventRecord(e0, (hipStream_t)stream);
            for (int j = 0; j <= k; j++) hipLaunchKernelGGL(bo_k_nothing, dim3(1), dim3(64), 0, (hipStream_t)stream);
            if (!rc) rc = (int)hipEventRecord(e1, (hipStream_t)stream);
            if (!rc) rc = (int)hipEventSynchronize(e1);
            float ms = 0.0f;
            if (!rc) rc = (int)hipEventElapsedTime(&ms, e0, e1);
            if (!rc && i >= 4) v[k].push_back(ms);
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    RT(rc);
    std::sort(v[0].begin(), v[0].end());
    std::sort(v[1].begin(), v[1].end());
    const double p1 = v[0][v[0].size() / 2], p2 = v[1][v[1].size() / 2], ov = 2.0 * p1 - p2;
    *ms_out = ov > 0.0 ? ov : 0.0;
    return BO_OK;
#endif
}

extern "C" int bo_fast_stats(bo_engine *e, uint64_t *granules_read, uint64_t *path_nodes, int32_t *arena_granules, int32_t time_select, double *select_ms,
                             int64_t *select_launches, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    if (!e->fast) return fail(BO_E_CONFIG, "bo_fast_stats: fast-mode engines only");
    const size_t G = (size_t)e->d.c.G;
    std::vector<int32_t> tmp(2 * G);
    const size_t pitch = (size_t)e->f.CS * 4;
    if (granules_read) RT(rt_d2h_2d(tmp.data(), 4, e->f.ctl + FWC_GRAN, pitch, 4, G, stream));
    if (path_nodes) RT(rt_d2h_2d(tmp.data() + G, 4, e->f.ctl + FWC_PNODES, pitch, 4, G, stream));
    if (arena_granules) RT(rt_d2h_2d(arena_granules, 4, e->f.ctl + FWC_TOP, pitch, 4, G, stream));
    RT(rt_sync(stream));
    for (size_t g = 0; g < G; g++) {  // (32-bit counters per game on the device: 2^31 granules are 256 GB through one game's select path)
        if (granules_read) granules_read[g] = (uint64_t)(uint32_t)tmp[g];
        if (path_nodes) path_nodes[g] = (uint64_t)(uint32_t)tmp[G + g];
    }
    RT(rt_sync(stream));
#if !defined(BO_WAVE_EMU)
    if (e->sel_pending) {
        float ms = 0.0f;
        RT((int)hipEventSynchronize(e->sel_ev1));
        RT((int)hipEventElapsedTime(&ms, e->sel_ev0, e->sel_ev1));
        e->sel_ms += ms; e->sel_launches++; e->sel_pending = 0;
    }
#endif
    if (select_ms) *select_ms = e->sel_ms;
    if (select_launches) *select_launches = e->sel_launches;
    if (time_select >= 0) {
        if (time_select && !e->sel_profile) { e->sel_ms = 0.0; e->sel_launches = 0; }
        e->sel_profile = time_select ? 1 : 0;
    }
    return BO_OK;
}

extern "C" int bo_engine_status(bo_engine *e, int32_t *status, int32_t *evals, int32_t *flushes, int32_t *term_sims,
                                int32_t *levels, int32_t *children_scanned, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    const size_t G = (size_t)e->d.c.G;
    if (status) RT(rt_d2h(status, e->d.status, G * 4, stream));
    if (evals) RT(rt_d2h(evals, e->d.stat_evals, G * 4, stream));
    if (flushes) RT(rt_d2h(flushes, e->d.stat_flushes, G * 4, stream));
    if (e->fast) {  // the fast mode keeps these counters in the games' control blocks (bo_fastw.h)
        const size_t pitch = (size_t)e->f.CS * 4;
        if (term_sims) RT(rt_d2h_2d(term_sims, 4, e->f.ctl + FWC_TERM, pitch, 4, G, stream));
        if (levels) RT(rt_d2h_2d(levels, 4, e->f.ctl + FWC_LEVELS, pitch, 4, G, stream));
        if (children_scanned) RT(rt_d2h_2d(children_scanned, 4, e->f.ctl + FWC_KIDS, pitch, 4, G, stream));
    } else {
        if (term_sims) RT(rt_d2h(term_sims, e->d.stat_term_sims, G * 4, stream));
        if (levels) RT(rt_d2h(levels, e->d.stat_levels, G * 4, stream));
        if (children_scanned) RT(rt_d2h(children_scanned, e->d.stat_children_scanned, G * 4, stream));
    }
    RT(rt_sync(stream));
    return BO_OK;
}

extern "C" int bo_debug_profile(bo_engine *e, int enable, uint64_t *cycles_out, void *stream) {
    if (!e) return fail(BO_E_ARG, "null engine");
    const size_t G = (size_t)e->d.c.G;
    if (cycles_out) {
        RT(rt_d2h(cycles_out, e->d.prof, G * BO_PROF_SLOTS * sizeof(uint64_t), stream));
        RT(rt_sync(stream));
    }
    if (enable >= 0) {
        if (enable && !e->d.c.profile) RT(rt_memset(e->d.prof, 0, G * BO_PROF_SLOTS * sizeof(uint64_t), stream));
        e->d.c.profile = enable;  // 1: every game-step; N > 1: only game-steps longer than N cycles
    }
    return BO_OK;
}

extern "C" int bo_movegen_batch(bo_engine *e, int n, const bo_position *pos, int32_t *moves_out, int32_t *n_out,
                                int32_t *check_out, void *stream) {
    if (!e || n < 1 || !pos || !moves_out || !n_out) return fail(BO_E_ARG, "bad arguments");
    std::vector<DPos> hp((size_t)n);
    for (int i = 0; i < n; i++) hp[i] = from_abi(pos[i]);
    void *dp = nullptr, *dm = nullptr, *dn = nullptr, *dc = nullptr;
    int rc = rt_malloc(&dp, hp.size() * sizeof(DPos)) | rt_malloc(&dm, (size_t)n * BO_MAX_MOVES * sizeof(bo_mv)) |
             rt_malloc(&dn, (size_t)n * 4) | rt_malloc(&dc, (size_t)n * 4);
    std::vector<bo_mv> hm((size_t)n * BO_MAX_MOVES);
    std::vector<int> hc((size_t)n);
    if (!rc) rc = rt_h2d(dp, hp.data(), hp.size() * sizeof(DPos), stream);
    if (!rc) rc = RT_LAUNCH(bo_k_movegen, n, stream, (const DPos *)dp, (bo_mv *)dm, (int *)dn, (int *)dc);
    if (!rc) rc = rt_d2h(hm.data(), dm, hm.size() * sizeof(bo_mv), stream) | rt_d2h(n_out, dn, (size_t)n * 4, stream) |
                  rt_d2h(hc.data(), dc, (size_t)n * 4, stream);
    int rc2 = rt_sync(stream);
    rt_free(dp); rt_free(dm); rt_free(dn); rt_free(dc);
    if (rc || rc2) return fail(BO_E_HIP, std::string("bo_movegen_batch: ") + rt_errstr(rc ? rc : rc2));
    for (size_t i = 0; i < hm.size(); i++) moves_out[i] = hm[i];
    if (check_out) for (int i = 0; i < n; i++) check_out[i] = hc[i];
    return BO_OK;
}

// ---- wide PUCT select (roofline workload; independent of an engine instance) ----------------------------------
extern "C" int bo_select_wide(const void *blocks_dev, const int32_t *root_block_dev, const int32_t *root_n_dev,
                              const float *sqrt_lut_dev, int n_trees, int max_depth, float cpuct, int grid_blocks,
                              int32_t *out_leaf_dev, int32_t *out_levels_dev, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)blocks_dev; (void)root_block_dev; (void)root_n_dev; (void)sqrt_lut_dev; (void)n_trees; (void)max_depth;
    (void)cpuct; (void)grid_blocks; (void)out_leaf_dev; (void)out_levels_dev; (void)stream;
    return fail(BO_E_CONFIG, "bo_select_wide is a gfx950-only kernel");
#else
    if (!blocks_dev || !root_block_dev || !root_n_dev || !sqrt_lut_dev || !out_leaf_dev || !out_levels_dev || n_trees < 1)
        return fail(BO_E_ARG, "bad arguments");
    const int U = 4;  // trees in flight per half-wave (bo_select_wide.h)
    if (grid_blocks < 1) {
        // one workgroup per 8*U trees and no grid-stride tail; measured on MI355X at 262 144 trees: 4096 (strided) 85.9 us,
        // 8192 (exact) 83.1 us, 16384 (half of the workgroups exit at once) 81.3 us -- launch twice the exact grid.
        grid_blocks = 2 * ((n_trees + 8 * U - 1) / (8 * U));
    }
    if (grid_blocks > 65536) grid_blocks = 65536;
    hipLaunchKernelGGL(bo_k_select_wide, dim3((unsigned)grid_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const WideBlock *)blocks_dev, (const int *)root_block_dev, (const int *)root_n_dev, sqrt_lut_dev,
                       n_trees, max_depth, cpuct, (int *)out_leaf_dev, (int *)out_levels_dev);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// ---- fused epilogues of the evaluate stage (bo_nn_fused.h); independent of an engine instance -----------------------
extern "C" int bo_nn_bias_act(float *x_dev, const float *bias_dev, const float *residual_dev, int batch, int channels,
                              void *stream) {
#if defined(BO_WAVE_EMU)
    (void)x_dev; (void)bias_dev; (void)residual_dev; (void)batch; (void)channels; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_bias_act is a gfx950-only kernel");
#else
    if (!x_dev || !bias_dev || batch < 1 || channels < 1) return fail(BO_E_ARG, "bad arguments");
    const long n4 = (long)batch * channels * 16;
    long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bo_k_bias_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_dev, bias_dev, residual_dev, n4,
                       channels);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

extern "C" int bo_nn_se_residual(float *x_dev, const float *bias_dev, const float *w1_dev, const float *w2_dev,
                                 const float *residual_dev, int batch, int channels, int hidden, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)x_dev; (void)bias_dev; (void)w1_dev; (void)w2_dev; (void)residual_dev; (void)batch; (void)channels; (void)hidden; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_se_residual is a gfx950-only kernel");
#else
    if (!x_dev || !bias_dev || !w1_dev || !w2_dev || !residual_dev || batch < 1) return fail(BO_E_ARG, "bad arguments");
    if (channels < 1 || channels > BO_SE_MAX_C || hidden < 1 || hidden > BO_SE_MAX_H) return fail(BO_E_CONFIG, "SE block: channels <= 256, hidden <= 32");
    const size_t lds = (size_t)channels * 65 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)bo_k_se_residual, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 65 * 4);
        attr_set = true;
    }
    hipLaunchKernelGGL(bo_k_se_residual, dim3((unsigned)batch), dim3(256), lds, (hipStream_t)stream, x_dev, bias_dev, w1_dev, w2_dev,
                       residual_dev, channels, hidden);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// small-batch form: the result replaces `residual_inout_dev`, x is only read (bo_nn_fused.h: bo_k_se_residual_small)
extern "C" int bo_nn_se_residual_small(const float *x_dev, const float *bias_dev, const float *w1_dev, const float *w2_dev,
                                       float *residual_inout_dev, int batch, int channels, int hidden, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)x_dev; (void)bias_dev; (void)w1_dev; (void)w2_dev; (void)residual_inout_dev; (void)batch; (void)channels; (void)hidden; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_se_residual_small is a gfx950-only kernel");
#else
    if (!x_dev || !bias_dev || !w1_dev || !w2_dev || !residual_inout_dev || batch < 1 || batch > 65535) return fail(BO_E_ARG, "bad arguments");
    if (channels < 16 || channels > BO_SE_MAX_C || (channels & 15) || hidden < 1 || hidden > 16)
        return fail(BO_E_CONFIG, "small SE block: channels a multiple of 16 <= 256, hidden <= 16");
    hipLaunchKernelGGL(bo_k_se_residual_small, dim3((unsigned)(channels / 16), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, x_dev, bias_dev,
                       w1_dev, w2_dev, residual_inout_dev, channels, hidden);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// ---- direct 3x3 convolution on the fp32 matrix cores (bo_conv.h); independent of an engine instance ---------------
extern "C" int bo_nn_conv3x3(const float *x_dev, const float *wpacked_dev, const float *bias_dev, const float *residual_dev,
                             float *y_dev, int batch, int c_in, int c_out, int mode, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)x_dev; (void)wpacked_dev; (void)bias_dev; (void)residual_dev; (void)y_dev; (void)batch; (void)c_in; (void)c_out; (void)mode; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_conv3x3 is a gfx950-only kernel");
#else
    if (!x_dev || !wpacked_dev || !bias_dev || !y_dev || batch < 1) return fail(BO_E_ARG, "bad arguments");
    if (mode < 0 || mode > 2 || (mode == 2 && !residual_dev)) return fail(BO_E_ARG, "bad epilogue mode");
    const bo_f32x4 *wp = reinterpret_cast<const bo_f32x4 *>(wpacked_dev);
    hipStream_t st = (hipStream_t)stream;
#define BO_CONV_CASE(CI, CO)                                                                                              \
    if (c_in == CI && c_out == CO) {                                                                                      \
        hipLaunchKernelGGL((bo_k_conv3x3<CI, CO>), dim3((unsigned)batch), dim3(CO * 2), 0, st, x_dev, wp, bias_dev, residual_dev, \
                           y_dev, mode);                                                                                  \
        RT((int)hipGetLastError());                                                                                       \
        return BO_OK;                                                                                                     \
    }
    BO_CONV_CASE(120, 64) BO_CONV_CASE(64, 64) BO_CONV_CASE(120, 128) BO_CONV_CASE(128, 128) BO_CONV_CASE(120, 256) BO_CONV_CASE(256, 256)
#undef BO_CONV_CASE
    return fail(BO_E_CONFIG, "bo_nn_conv3x3: supported (c_in, c_out): (120|C, C) for C in {64, 128, 256}");
#endif
}

// ---- the whole residual tower as one persistent kernel (bo_tower.h) ------------------------------------------------
#if !defined(BO_WAVE_EMU)
struct bo_tower_s {
    int channels = 0, n_layers = 0, n_cu = 0, device = 0, algo = 0;
    int head_channels = 0, head_split = 0, head_w_off = 0, head_b_off = 0;
    bo_f32x4 *wts = nullptr;
    float *params = nullptr;
    bo_tower_layer *layers = nullptr;
    int *overflow = nullptr;  // BO_TOWER_SPLIT_F16: an activation left the fp16 range
};
#else
struct bo_tower_s { int unused; };
#endif

extern "C" int bo_nn_tower_create(const bo_tower_layer_desc *layers, int n_layers, const float *weights, int64_t n_weights,
                                  const float *params, int64_t n_params, int channels, int algo, const bo_tower_head_desc *head, int device,
                                  bo_tower **out) {
#if defined(BO_WAVE_EMU)
    (void)head; (void)algo; (void)layers; (void)n_layers; (void)weights; (void)n_weights; (void)params; (void)n_params; (void)channels; (void)device; (void)out;
    return fail(BO_E_CONFIG, "bo_nn_tower is a gfx950-only kernel");
#else
    static_assert(sizeof(bo_tower_layer_desc) == sizeof(bo_tower_layer), "descriptor layouts must agree");
    if (!layers || !weights || !params || !out || n_layers < 1 || n_layers > 4096) return fail(BO_E_ARG, "bad arguments");
    const bool split_w = algo == BO_TOWER_SPLIT_F16 || algo == BO_TOWER_SPLIT_F16_T16;
    const bool f16_w = algo == BO_TOWER_DIRECT_F16 || algo == BO_TOWER_DIRECT_F16_T16;
    const bool half_w = f16_w || split_w;  // fp16 weight fragments, 16-byte offsets
    if (half_w ? (channels != 128 && channels != 256) : (channels != 64 && channels != 128))
        return fail(BO_E_CONFIG, "bo_nn_tower: channels must be 64 or 128 (fp32: two padded images per board in LDS) or 128 / 256 (BO_TOWER_DIRECT_F16, BO_TOWER_SPLIT_F16)");
    if (n_weights % 4) return fail(BO_E_ARG, "n_weights must be a multiple of 4");
    if (algo != BO_TOWER_DIRECT && algo != BO_TOWER_WINOGRAD && !half_w) return fail(BO_E_ARG, "unknown algo");
    if (algo == BO_TOWER_SPLIT_F16_T16 && channels != 128) return fail(BO_E_CONFIG, "BO_TOWER_SPLIT_F16_T16: 128 filters (other widths: BO_TOWER_SPLIT_F16)");
    if (f16_w && !head) return fail(BO_E_ARG, "BO_TOWER_DIRECT_F16 needs the fused head (it has no tower output buffer)");
    const int C = channels;
    const int split = split_w ? 2 : 1;  // (hi, lo) fragment pairs; one more float (the inverse weight scale) behind every bias
    for (int l = 0; l < n_layers; l++) {  // every offset the kernel will form stays inside the two buffers
        const bo_tower_layer_desc &L = layers[l];
        // K steps per layer: direct = groups of 8 input channels, Winograd = groups of 4; the input conv is padded to 128
        const int cin = L.kind == 0 ? 128 : C;
        const int want_t4 = algo == BO_TOWER_DIRECT ? cin / 8 : algo == BO_TOWER_WINOGRAD ? cin / 4 : 9 * cin / 16;
        const int64_t w4 = algo == BO_TOWER_DIRECT ? (int64_t)9 * want_t4 * C * 2
                           : algo == BO_TOWER_WINOGRAD ? (int64_t)want_t4 * (C / 16) * 4 * 64 : (int64_t)want_t4 * (C / 32) * 64 * split;
        if (L.kind < 0 || L.kind > 3 || (l == 0) != (L.kind == 0)) return fail(BO_E_ARG, "layer " + std::to_string(l) + ": bad kind");
        if (L.kind == 1 && (l + 1 >= n_layers || layers[l + 1].kind < 2)) return fail(BO_E_ARG, "a first conv must be followed by a second conv");
        if (L.kind >= 2 && layers[l - 1].kind != 1) return fail(BO_E_ARG, "a second conv must follow a first conv");
        if (L.t4 != want_t4) return fail(BO_E_ARG, "layer " + std::to_string(l) + ": t4 must be " + std::to_string(want_t4));
        if (L.w_off4 < 0 || ((int64_t)L.w_off4 + w4) * 4 > n_weights) return fail(BO_E_ARG, "weights offset out of range");
        if (L.bias_off < 0 || (int64_t)L.bias_off + C + (split - 1) > n_params) return fail(BO_E_ARG, "bias offset out of range");
        if ((algo == BO_TOWER_WINOGRAD || split_w) && (L.bias_off & 3)) return fail(BO_E_ARG, "BO_TOWER_WINOGRAD / BO_TOWER_SPLIT_F16: bias_off must be a multiple of 4 floats");
        if (L.kind == 3) {
            if (L.hidden < 1 || L.hidden > 16) return fail(BO_E_CONFIG, "SE hidden width must be 1..16");
            if (half_w && L.hidden > C / 16) return fail(BO_E_CONFIG, "fp16-pipe towers: SE hidden width must be <= channels/16");
            if (L.se_w1_off < 0 || (int64_t)L.se_w1_off + (int64_t)L.hidden * C > n_params || L.se_w2_off < 0 ||
                (int64_t)L.se_w2_off + (int64_t)L.hidden * C > n_params)
                return fail(BO_E_ARG, "SE weight offset out of range");
        }
        if ((L.last != 0) != (l == n_layers - 1)) return fail(BO_E_ARG, "exactly the final layer stores the output");
    }
    if (layers[n_layers - 1].kind < 2) return fail(BO_E_ARG, "the tower must end with a second conv");
    if (head) {
        if (algo == BO_TOWER_DIRECT) return fail(BO_E_CONFIG, "fused head convolutions need BO_TOWER_WINOGRAD, BO_TOWER_DIRECT_F16 or BO_TOWER_SPLIT_F16");
        if (head->channels < 1 || head->channels > 256 || head->split < 0 || head->split > head->channels) return fail(BO_E_ARG, "bad head channels/split");
        if (head->b_off < 0 || (int64_t)head->b_off + head->channels + (split - 1) > n_params) return fail(BO_E_ARG, "head bias offset out of range");
        if (algo == BO_TOWER_WINOGRAD) {
            if (head->w_off < 0 || (head->w_off & 3) || (int64_t)head->w_off + (int64_t)((head->channels + 15) / 16) * 16 * C > n_params)
                return fail(BO_E_ARG, "head weight offset out of range");
        } else if (head->w_off < 0 || ((int64_t)head->w_off + (int64_t)((head->channels + 31) / 32) * (C / 16) * 64 * split) * 4 > n_weights) {
            return fail(BO_E_ARG, "head weight offset out of range");
        }
    }
    RT(rt_set_device(device));
    hipDeviceProp_t prop;
    RT((int)hipGetDeviceProperties(&prop, device));
    bo_tower_s *t = new bo_tower_s();
    t->channels = C; t->n_layers = n_layers; t->n_cu = prop.multiProcessorCount; t->device = device; t->algo = algo;
    if (head) { t->head_channels = head->channels; t->head_split = head->split; t->head_w_off = head->w_off; t->head_b_off = head->b_off; }
    int rc = (int)hipMalloc((void **)&t->wts, (size_t)n_weights * 4);
    if (!rc) rc = (int)hipMalloc((void **)&t->params, (size_t)n_params * 4);
    if (!rc) rc = (int)hipMalloc((void **)&t->layers, (size_t)n_layers * sizeof(bo_tower_layer));
    if (!rc) rc = (int)hipMalloc((void **)&t->overflow, 8);  // [status word | scratch of bo_nn_tower_status's exchange]
    if (!rc) rc = (int)hipMemset(t->overflow, 0, 8);
    if (!rc) rc = (int)hipMemcpy(t->wts, weights, (size_t)n_weights * 4, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMemcpy(t->params, params, (size_t)n_params * 4, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMemcpy(t->layers, layers, (size_t)n_layers * sizeof(bo_tower_layer), hipMemcpyHostToDevice);
    if (rc) {
        (void)hipFree(t->wts); (void)hipFree(t->params); (void)hipFree(t->layers); (void)hipFree(t->overflow);
        delete t;
        return fail(BO_E_HIP, std::string("bo_nn_tower_create: ") + rt_errstr(rc));
    }
    *out = t;
    return BO_OK;
#endif
}

// Rate (kHz) of the constant-rate clock the in-kernel timings are taken with (wall_clock64: hipDeviceAttributeWallClockRate).
extern "C" int bo_device_wall_clock_khz(int device, int32_t *khz_out) {
#if defined(BO_WAVE_EMU)
    (void)device; (void)khz_out;
    return fail(BO_E_CONFIG, "gfx950 only");
#else
    if (!khz_out) return fail(BO_E_ARG, "null argument");
    int v = 0;
    RT((int)hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, device));
    *khz_out = v;
    return BO_OK;
#endif
}

static int tower_forward_impl(bo_tower *t, const float *x_dev, float *y_dev, void *head_a_dev, void *head_b_dev, int batch, void *timing_dev, void *stream);
extern "C" int bo_nn_tower_forward(bo_tower *t, const float *x_dev, float *y_dev, void *head_a_dev, void *head_b_dev, int batch,
                                   void *stream) {
    return tower_forward_impl(t, x_dev, y_dev, head_a_dev, head_b_dev, batch, nullptr, stream);
}
// The same launch with its duration noted by the kernel itself in `timing_dev` (BO_TOWER_SPLIT_F16 only): uint64 [seq | arrivals |
// start[4096] | end[4096]], zeroed by the caller; launch k made with this buffer leaves its first workgroup's start and its last
// workgroup's end in slot k % 4096 (units: the device's constant-rate clock, hipDeviceAttributeWallClockRate kHz).  One launch per
// buffer at a time (launches of one stream).  For measurements inside captured graphs, where no event pair fits between two nodes.
extern "C" int bo_nn_tower_forward_timed(bo_tower *t, const float *x_dev, float *y_dev, void *head_a_dev, void *head_b_dev, int batch,
                                         void *timing_dev, void *stream) {
    return tower_forward_impl(t, x_dev, y_dev, head_a_dev, head_b_dev, batch, timing_dev, stream);
}
static int tower_forward_impl(bo_tower *t, const float *x_dev, float *y_dev, void *head_a_dev, void *head_b_dev, int batch, void *timing_dev, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)x_dev; (void)y_dev; (void)head_a_dev; (void)head_b_dev; (void)batch; (void)stream; (void)timing_dev;
    return fail(BO_E_CONFIG, "bo_nn_tower is a gfx950-only kernel");
#else
    if (!t || !x_dev || batch < 1) return fail(BO_E_ARG, "bad arguments");
    bo_tower_head hd;
    if (t->head_channels > 0) {
        if ((t->head_split > 0 && !head_a_dev) || (t->head_split < t->head_channels && !head_b_dev)) return fail(BO_E_ARG, "head output buffers missing");
        hd.channels = t->head_channels; hd.split = t->head_split; hd.w_off = t->head_w_off; hd.b_off = t->head_b_off;
        hd.out_a = (float *)head_a_dev; hd.out_b = (float *)head_b_dev;
    } else if (!y_dev) {
        return fail(BO_E_ARG, "y_dev is required for a tower without fused heads");
    }
    const int slots = t->n_cu * (t->channels == 64 && t->algo == BO_TOWER_DIRECT ? 2 : 1);  // direct, 64 filters: two 2-wave workgroups share a CU
    const unsigned grid = (unsigned)(batch < slots ? batch : slots);
    hipStream_t st = (hipStream_t)stream;
    if (t->algo == BO_TOWER_DIRECT_F16 || t->algo == BO_TOWER_DIRECT_F16_T16) {  // two boards per workgroup
        bo_tower_head_h hh;
        hh.channels = t->head_channels; hh.split = t->head_split; hh.w_off8 = t->head_w_off; hh.b_off = t->head_b_off;
        hh.out_a = (_Float16 *)head_a_dev; hh.out_b = (_Float16 *)head_b_dev;
        const int pairs = (batch + 1) / 2;
        const unsigned g2 = (unsigned)(pairs < t->n_cu ? pairs : t->n_cu);
        const bo_h8 *w8 = reinterpret_cast<const bo_h8 *>(t->wts);
        static const int ar_h16 = [] { const char *v = getenv("BETAONE_TOWER_H16_AR"); return (v && atoi(v) == 6) ? 6 : 3; }();  // (LAB: ring depth of the 256-filter instance)
        if (t->algo == BO_TOWER_DIRECT_F16_T16 && t->channels == 256 && ar_h16 == 6) hipLaunchKernelGGL((bo_k_tower_h16<256, 2, 6>), dim3(g2), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, batch, hh);
        else if (t->algo == BO_TOWER_DIRECT_F16_T16 && t->channels == 256) hipLaunchKernelGGL((bo_k_tower_h16<256, 2, 3>), dim3(g2), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, batch, hh);
        else if (t->algo == BO_TOWER_DIRECT_F16_T16) hipLaunchKernelGGL((bo_k_tower_h16<128, 1>), dim3(g2), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, batch, hh);
        else if (t->channels == 256) hipLaunchKernelGGL((bo_k_tower_h<256, 2>), dim3(g2), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, batch, hh);
        else hipLaunchKernelGGL((bo_k_tower_h<128, 1>), dim3(g2), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, batch, hh);
        RT((int)hipGetLastError());
        return BO_OK;
    }
    if (t->algo == BO_TOWER_SPLIT_F16 || t->algo == BO_TOWER_SPLIT_F16_T16) {  // float32 in and out, fp16 (hi, lo) pairs on the matrix pipe; one board per workgroup
        bo_tower_head_s hs;
        hs.channels = t->head_channels; hs.split = t->head_split; hs.w_off8 = t->head_w_off; hs.b_off = t->head_b_off;
        hs.out_a = (float *)head_a_dev; hs.out_b = (float *)head_b_dev; hs.overflow = t->overflow;
        hs.timing = (unsigned long long *)timing_dev;
        const bo_h8 *w8 = reinterpret_cast<const bo_h8 *>(t->wts);
        // 256 filters: two tiles per wave double every register set.  Weight fragments 8 K-steps ahead: 41 spilled registers (160 B of
        // scratch); 4 ahead: none, and as fast (1 667-1 672 us against 1 678-1 679 per 256 boards, same box: profiles/r04_tower256_ring.md).
        // BETAONE_TOWER256_AR=8 keeps the old instance selectable for A/B runs.
        static const int ar256 = [] { const char *v = getenv("BETAONE_TOWER256_AR"); return (v && v[0] == '8') ? 8 : 4; }();
        if (t->algo == BO_TOWER_SPLIT_F16_T16)  // the same products as 16x16x32 tiles (bo_tower_s16.h: the chip holds a higher clock under them)
            // (weight fragments 6 K-steps = 24 KiB per wave ahead; 12 ahead needs all 512 registers + 12 spilled and measured 185 us
            // against 179 for a lone 64-board launch: profiles/r05_device_turn_and_tiles.md)
            hipLaunchKernelGGL((bo_k_tower_s16<6>), dim3(grid), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, y_dev, batch, hs);
        else if (t->channels == 256 && ar256 == 8) hipLaunchKernelGGL((bo_k_tower_s<256, 2, 1, 8>), dim3(grid), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, y_dev, batch, hs);
        else if (t->channels == 256) hipLaunchKernelGGL((bo_k_tower_s<256, 2, 1, 4>), dim3(grid), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, y_dev, batch, hs);
        else  // (B operands read two K-steps ahead, weight fragments requested twelve ahead: profiles/r03_split_tower.md)
            hipLaunchKernelGGL((bo_k_tower_s<128, 1, 2, 12>), dim3(grid), dim3(256), 0, st, x_dev, w8, t->params, t->layers, t->n_layers, y_dev, batch, hs);
        RT((int)hipGetLastError());
        return BO_OK;
    }
    if (t->algo == BO_TOWER_WINOGRAD && t->channels == 128)
        hipLaunchKernelGGL((bo_k_tower_wg<128>), dim3(grid), dim3(512), 0, st, x_dev, t->wts, t->params, t->layers, t->n_layers, y_dev, batch, hd);
    else if (t->algo == BO_TOWER_WINOGRAD)
        hipLaunchKernelGGL((bo_k_tower_wg<64>), dim3(grid), dim3(256), 0, st, x_dev, t->wts, t->params, t->layers, t->n_layers, y_dev, batch, hd);
    else if (t->channels == 128)
        hipLaunchKernelGGL((bo_k_tower<128>), dim3(grid), dim3(256), 0, st, x_dev, t->wts, t->params, t->layers, t->n_layers, y_dev, batch);
    else
        hipLaunchKernelGGL((bo_k_tower<64>), dim3(grid), dim3(128), 0, st, x_dev, t->wts, t->params, t->layers, t->n_layers, y_dev, batch);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

extern "C" void bo_nn_tower_destroy(bo_tower *t) {
#if !defined(BO_WAVE_EMU)
    if (!t) return;
    (void)hipFree(t->wts); (void)hipFree(t->params); (void)hipFree(t->layers); (void)hipFree(t->overflow);
#endif
    delete t;
}

#if !defined(BO_WAVE_EMU)
__global__ void bo_k_word_exchange(int *word, int *out) { *out = atomicExch(word, 0); }
#endif

// Read AND clear the tower's status word in one atomic exchange on `stream` -- the stream the tower is launched on: ordered behind
// the forwards it reports on, and a forward that sets the word while it is being read is not lost between a copy and a memset.
extern "C" int bo_nn_tower_status(bo_tower *t, int32_t *overflow_out, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)overflow_out; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_tower is a gfx950-only kernel");
#else
    if (!t || !overflow_out) return fail(BO_E_ARG, "bad arguments");
    hipLaunchKernelGGL(bo_k_word_exchange, dim3(1), dim3(1), 0, (hipStream_t)stream, t->overflow, t->overflow + 1);
    RT((int)hipGetLastError());
    RT(rt_d2h(overflow_out, t->overflow + 1, 4, stream));
    RT(rt_sync(stream));
    return BO_OK;
#endif
}

// Device address of the tower's status word (non-zero: an activation left the fp16 range), for bo_engine_watch.
extern "C" int bo_nn_tower_word(bo_tower *t, void **dev_word_out) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)dev_word_out;
    return fail(BO_E_CONFIG, "bo_nn_tower is a gfx950-only kernel");
#else
    if (!t || !dev_word_out) return fail(BO_E_ARG, "bad arguments");
    *dev_word_out = t->overflow;
    return BO_OK;
#endif
}

// ---- the tower of one board (a few boards) as ONE launch spread over the chip (bo_tower_b1.h): uci.py's batch-1 evaluations ------
struct bo_b1_s {
#if !defined(BO_WAVE_EMU)
    int device = 0, C = 0, n_layers = 0, max_batch = 0;
    bool split = false;             // every layer came with (hi, lo) fp16 weights: the tiles multiply on the fp16 matrix pipe
    bo_b1_layer *layers = nullptr;  // device table
    float *bufs = nullptr, *pool = nullptr;
    unsigned *sync = nullptr;       // [max_batch counters | status word | padding], a block of its own (zeroed before every launch)
    size_t sync_bytes = 0;
    unsigned long long *prof = nullptr;  // bo_nn_b1_profile: per-wave phase clocks of the next launches (NULL: off)
#endif
};

extern "C" int bo_nn_b1_create(const bo_b1_layer_desc *layers, int n_layers, int channels, int max_batch, int device, bo_b1 **out) {
#if defined(BO_WAVE_EMU)
    (void)layers; (void)n_layers; (void)channels; (void)max_batch; (void)device; (void)out;
    return fail(BO_E_CONFIG, "bo_nn_b1 is a gfx950-only kernel");
#else
    if (!layers || !out || n_layers < 1 || n_layers > BO_B1_MAX_LAYERS || (n_layers & 1) == 0) return fail(BO_E_ARG, "bo_nn_b1_create: 1 + 2 x blocks layers");
    if (channels != 64 && channels != 128 && channels != 256) return fail(BO_E_CONFIG, "bo_nn_b1: 64, 128 or 256 filters");
    const int tiles = (channels / 16) * 4;
    if (max_batch < 1 || max_batch * tiles > 256) return fail(BO_E_CONFIG, "bo_nn_b1: (filters / 16) x 4 x batch workgroups must be resident at once (<= 256)");
    std::vector<bo_b1_layer> tab((size_t)n_layers);
    int n_split = 0;
    for (int l = 0; l < n_layers; l++) {
        const bo_b1_layer_desc &d = layers[l];
        if (!d.weights_dev || !d.bias_dev) return fail(BO_E_ARG, "bo_nn_b1_create: null weights");
        const int want_cin = l == 0 ? 128 : channels, want_mode_lo = (l == 0 || (l & 1)) ? 0 : 1;
        if (d.c_in != want_cin || d.c_in_x < 1 || d.c_in_x > d.c_in || (l > 0 && d.c_in_x != channels)) return fail(BO_E_ARG, "bo_nn_b1_create: layer channel counts");
        if ((l == 0 || (l & 1)) ? d.mode != 0 : (d.mode != 1 && d.mode != 2)) return fail(BO_E_ARG, "bo_nn_b1_create: layer modes must be 0, then (0, 1 | 2) per block");
        (void)want_mode_lo;
        if (d.mode == 2 && (!d.se_w1_dev || !d.se_w2_dev || d.se_hidden < 1 || d.se_hidden > 16)) return fail(BO_E_CONFIG, "bo_nn_b1: SE hidden width 1..16");
        tab[l].w = (const bo_f32x4 *)d.weights_dev; tab[l].bias = d.bias_dev; tab[l].se_w1 = d.se_w1_dev; tab[l].se_w2 = d.se_w2_dev;
        tab[l].cin = d.c_in; tab[l].cin_x = d.c_in_x; tab[l].mode = d.mode; tab[l].se_h = d.se_hidden;
        tab[l].w_split = (const bo_f32x4 *)d.weights_split_dev; tab[l].inv_scale = d.inv_scale; tab[l].pad = 0;
        if (d.weights_split_dev && !(d.inv_scale > 0.0f)) return fail(BO_E_ARG, "bo_nn_b1_create: split weights need inv_scale > 0");
        n_split += d.weights_split_dev ? 1 : 0;
    }
    if (n_split != 0 && n_split != n_layers) return fail(BO_E_ARG, "bo_nn_b1_create: split weights for every layer or for none");
    RT((int)hipSetDevice(device));
    bo_b1 *t = new bo_b1();
    t->device = device; t->C = channels; t->n_layers = n_layers; t->max_batch = max_batch; t->split = n_split == n_layers;
    t->sync_bytes = (((size_t)2 * max_batch + 2) * 4 + 63) / 64 * 64;
    int rc = (int)hipMalloc((void **)&t->layers, tab.size() * sizeof(bo_b1_layer));
    if (!rc) rc = (int)hipMemcpy(t->layers, tab.data(), tab.size() * sizeof(bo_b1_layer), hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMalloc((void **)&t->bufs, (size_t)3 * max_batch * channels * 64 * 4);
    if (!rc) rc = (int)hipMalloc((void **)&t->pool, (size_t)max_batch * 4 * channels * 4);
    if (!rc) rc = (int)hipMalloc((void **)&t->sync, t->sync_bytes);
    if (!rc) rc = (int)hipMemset(t->sync, 0, t->sync_bytes);
    if (rc) {
        (void)hipFree(t->layers); (void)hipFree(t->bufs); (void)hipFree(t->pool); (void)hipFree(t->sync);
        delete t;
        return fail(BO_E_HIP, std::string("bo_nn_b1_create: ") + hipGetErrorString((hipError_t)rc));
    }
    *out = t;
    return BO_OK;
#endif
}

// One evaluation's tower: x [batch,120,8,8] -> y [batch,C,8,8] on `stream`.  The handle's buffers are used by the launch: one launch
// of a handle at a time (launches on ONE stream are in order; two streams need two handles).  Capturable (memset node + kernel node).
extern "C" int bo_nn_b1_forward(bo_b1 *t, const float *x_dev, float *y_dev, int batch, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)x_dev; (void)y_dev; (void)batch; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_b1 is a gfx950-only kernel");
#else
    if (!t || !x_dev || !y_dev) return fail(BO_E_ARG, "null argument");
    if (batch < 1 || batch > t->max_batch) return fail(BO_E_ARG, "bo_nn_b1_forward: batch beyond the handle's max_batch");
    hipStream_t st = (hipStream_t)stream;
    // No memset in front of the launch: the arrival counters run on from launch to launch (bo_b1_args::sync).
    bo_b1_args a;
    a.x = x_dev; a.y = y_dev; a.bufs = t->bufs; a.pool = t->pool; a.sync = t->sync; a.layers = t->layers; a.n_layers = t->n_layers; a.B = batch;
    a.prof = t->prof; a.MB = t->max_batch;
    const dim3 grid((unsigned)((t->C / 16) * 4), (unsigned)batch);
    static const int mode = [] { const char *v = getenv("BETAONE_B1_MODE"); return (v && v[0] >= '0' && v[0] <= '2') ? v[0] - '0' : (int)BO_B1_SC1; }();
#define BO_B1_LAUNCH_P(CC, PP)                                                                                       \
    do {                                                                                                             \
        if (mode == BO_B1_ACQ) hipLaunchKernelGGL((bo_k_tower_b1<CC, BO_B1_ACQ, PP>), grid, dim3(256), 0, st, a);    \
        else if (mode == BO_B1_NT) hipLaunchKernelGGL((bo_k_tower_b1<CC, BO_B1_NT, PP>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((bo_k_tower_b1<CC, BO_B1_SC1, PP>), grid, dim3(256), 0, st, a);                      \
    } while (0)
#define BO_B1_LAUNCH(CC) do { if (t->split) BO_B1_LAUNCH_P(CC, BO_B1_SPLIT); else BO_B1_LAUNCH_P(CC, BO_B1_F32); } while (0)
    if (t->C == 256) BO_B1_LAUNCH(256);
    else if (t->C == 128) BO_B1_LAUNCH(128);
    else BO_B1_LAUNCH(64);
#undef BO_B1_LAUNCH_P
#undef BO_B1_LAUNCH
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// *code_out = 0, or 1 + the phase of a hand-off wait that gave up in the LAST launch (the evaluation's output is then invalid).
// Synchronises `stream` (the stream the launches went to).
extern "C" int bo_nn_b1_status(bo_b1 *t, int32_t *code_out, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)code_out; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_b1 is a gfx950-only kernel");
#else
    if (!t || !code_out) return fail(BO_E_ARG, "bad arguments");
    int32_t two[2] = {0, 0};
    RT(rt_d2h(two, t->sync + 2 * t->max_batch, 8, stream));
    RT(rt_sync(stream));
    *code_out = two[0] ? two[0] : (two[1] ? -1 : 0);
    if (two[0] || two[1]) {  // a fault leaves the counters anywhere: start over from zero (the stream is idle here)
        RT((int)hipMemsetAsync(t->sync, 0, t->sync_bytes, (hipStream_t)stream));
        RT(rt_sync(stream));
    }
    return BO_OK;
#endif
}

// Device address of the handle's two status words [timeout code | saturation flag], for bo_engine_watch_words(.., 2): a search or
// a self-play ply then learns of a hand-off that gave up (or a saturated activation) with the result block it fetches anyway.
extern "C" int bo_nn_b1_word(bo_b1 *t, void **dev_words_out) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)dev_words_out;
    return fail(BO_E_CONFIG, "bo_nn_b1 is a gfx950-only kernel");
#else
    if (!t || !dev_words_out) return fail(BO_E_ARG, "bad arguments");
    *dev_words_out = t->sync + 2 * t->max_batch;
    return BO_OK;
#endif
}

// LAB: shader-clock sums per wave of the phases of a layer {wait for the hand-off, slab loads + staging, matrix pipe, reduction,
// epilogue + signal, layers} over the launches since enable = 1; enable = 0 copies them out ([batch * tiles * 4 waves][8], `cap`
// entries of 8) and switches the stamps off.  Synchronises the device.
extern "C" int bo_nn_b1_profile(bo_b1 *t, int enable, uint64_t *out, int cap) {
#if defined(BO_WAVE_EMU)
    (void)t; (void)enable; (void)out; (void)cap;
    return fail(BO_E_CONFIG, "bo_nn_b1 is a gfx950-only kernel");
#else
    if (!t) return fail(BO_E_ARG, "null handle");
    const size_t n = (size_t)t->max_batch * (size_t)(t->C / 16) * 4 * 4 * 8;
    RT((int)hipDeviceSynchronize());
    if (enable) {
        if (!t->prof) RT((int)hipMalloc((void **)&t->prof, n * 8));
        RT((int)hipMemset(t->prof, 0, n * 8));
        return BO_OK;
    }
    if (!t->prof) return fail(BO_E_STATE, "bo_nn_b1_profile: not enabled");
    if (out) RT((int)hipMemcpy(out, t->prof, std::min(n, (size_t)cap * 8) * 8, hipMemcpyDeviceToHost));
    (void)hipFree(t->prof);
    t->prof = nullptr;
    return BO_OK;
#endif
}

extern "C" void bo_nn_b1_destroy(bo_b1 *t) {
#if !defined(BO_WAVE_EMU)
    if (!t) return;
    (void)hipFree(t->prof);
    (void)hipFree(t->layers); (void)hipFree(t->bufs); (void)hipFree(t->pool); (void)hipFree(t->sync);
#endif
    delete t;
}

// LAB: a one-thread kernel that notes the device's constant-rate clock: ring[0] = entries written (atomic), entry k = ring[1 + 2k] =
// tag, ring[2 + 2k] = wall_clock64() -- enqueued between the launches of a stream (also inside captured graphs) it gives the timeline of
// that stream's phases as the device ran them (rocprofv3 changes how streams overlap; event pairs cannot sit between graph nodes).
#if !defined(BO_WAVE_EMU)
__global__ void bo_k_stamp(unsigned long long *ring, unsigned long long tag, unsigned long long cap) {
    const unsigned long long k = __hip_atomic_fetch_add(ring, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k < cap) { ring[1 + 2 * k] = tag; ring[2 + 2 * k] = wall_clock64(); }
}
#endif
extern "C" int bo_debug_stamp(void *ring_dev, uint64_t tag, uint64_t capacity, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)ring_dev; (void)tag; (void)capacity; (void)stream;
    return fail(BO_E_CONFIG, "gfx950 only");
#else
    if (!ring_dev) return fail(BO_E_ARG, "null ring");
    hipLaunchKernelGGL(bo_k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)ring_dev, (unsigned long long)tag, (unsigned long long)capacity);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// A stream whose kernels run on the compute units named by the mask only (bit i = CU i) -- and which has a hardware queue of its own, which
// is what K > 2 cohorts need first (pool streams share queues; profiles/r04_cohort_cu_masks.md: full masks 3.03 ms, pool streams 4.1 ms per ply).
extern "C" int bo_stream_create_cu_mask(int device, const uint32_t *mask_words, int n_words, void **stream_out) {
#if defined(BO_WAVE_EMU)
    (void)device; (void)mask_words; (void)n_words; (void)stream_out;
    return fail(BO_E_CONFIG, "gfx950 only");
#else
    if (!mask_words || n_words < 1 || !stream_out) return fail(BO_E_ARG, "bad arguments");
    bool any = false;
    for (int i = 0; i < n_words; i++) any = any || mask_words[i] != 0;
    if (!any) return fail(BO_E_ARG, "an empty CU mask would never run a kernel");
    RT((int)hipSetDevice(device));
    hipStream_t st = nullptr;
    RT((int)hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words));
    *stream_out = (void *)st;
    return BO_OK;
#endif
}
extern "C" int bo_stream_destroy(void *stream) {
#if defined(BO_WAVE_EMU)
    (void)stream;
    return fail(BO_E_CONFIG, "gfx950 only");
#else
    if (!stream) return BO_OK;
    RT((int)hipStreamDestroy((hipStream_t)stream));
    return BO_OK;
#endif
}

// ---- policy FC + softmax + value head behind the tower: two launches (bo_heads.h) ------------------------------------
extern "C" int bo_nn_heads(const void *p_dev, const void *v_dev, const float *wp_dev, const float *bp_dev, const float *w1_dev,
                           const float *b1_dev, const float *w2_dev, const float *b2_dev, float *policy_out_dev, float *value_out_dev,
                           float *scratch_dev, int batch, int flags, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)p_dev; (void)v_dev; (void)wp_dev; (void)bp_dev; (void)w1_dev; (void)b1_dev; (void)w2_dev; (void)b2_dev; (void)policy_out_dev;
    (void)value_out_dev; (void)scratch_dev; (void)batch; (void)flags; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_heads is a gfx950-only kernel");
#else
    if (!p_dev || !v_dev || !wp_dev || !bp_dev || !w1_dev || !b1_dev || !w2_dev || !b2_dev || !policy_out_dev || (!value_out_dev && !(flags & 4)) || !scratch_dev)
        return fail(BO_E_ARG, "null argument");
    if ((flags & 4) && (flags & 1)) return fail(BO_E_ARG, "bo_nn_heads: flags 4 (no rows kernel) leaves LOGITS: it excludes flags 1 (softmax)");
    if (batch < 1 || batch > 65536) return fail(BO_E_ARG, "bo_nn_heads: 1 <= batch <= 65536");
    bo_heads_args a;
    a.p = p_dev; a.v = v_dev; a.wp = wp_dev; a.bp = bp_dev; a.w1 = w1_dev; a.b1 = b1_dev; a.w2 = w2_dev; a.b2 = b2_dev;
    a.policy_out = policy_out_dev; a.value_out = value_out_dev;
    a.vpart = scratch_dev;  // [16 K chunks][batch][256]
    a.B = batch; a.softmax = flags & 1; a.pb = bo_heads_policy_boards(batch);
    const int otw = 4 / (a.pb >> 5);
    const unsigned tiles = (unsigned)(((batch + a.pb - 1) / a.pb) * ((BO_HEADS_NA / 32 + otw - 1) / otw) + ((batch + 63) / 64) * 4 * BO_HEADS_KS);
    if (flags & 2) hipLaunchKernelGGL(bo_k_heads_tiles<true>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);   // fp16 head planes
    else hipLaunchKernelGGL(bo_k_heads_tiles<false>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    if (!(flags & 4)) hipLaunchKernelGGL(bo_k_heads_rows, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, a);  // (4: bo_step_heads is the rows' consumer)
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// fp16 head planes and fp16 Linear weights, any number of rows: policy FC + softmax in one kernel, the whole value head in another
extern "C" int bo_nn_heads_f16(const void *p_dev, const void *v_dev, const void *wp_f16_dev, const float *bp_dev, const void *w1_f16_dev,
                               const float *b1_dev, const float *w2_dev, const float *b2_dev, float *policy_out_dev, float *value_out_dev,
                               float *scratch_dev, int batch, int flags, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)p_dev; (void)v_dev; (void)wp_f16_dev; (void)bp_dev; (void)w1_f16_dev; (void)b1_dev; (void)w2_dev; (void)b2_dev; (void)policy_out_dev;
    (void)value_out_dev; (void)scratch_dev; (void)batch; (void)flags; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_heads_f16 is a gfx950-only kernel");
#else
    if (!p_dev || !v_dev || !wp_f16_dev || !bp_dev || !w1_f16_dev || !b1_dev || !w2_dev || !b2_dev || !policy_out_dev || !value_out_dev)
        return fail(BO_E_ARG, "null argument");
    if (batch < 1 || batch > (1 << 22)) return fail(BO_E_ARG, "bo_nn_heads_f16: 1 <= batch <= 4194304");
    bo_heads_h_args a;
    a.p = (const _Float16 *)p_dev; a.v = (const _Float16 *)v_dev; a.wp = (const _Float16 *)wp_f16_dev; a.w1 = (const _Float16 *)w1_f16_dev;
    a.bp = bp_dev; a.b1 = b1_dev; a.w2 = w2_dev; a.b2 = b2_dev; a.policy_out = policy_out_dev; a.value_out = value_out_dev;
    a.B = batch; a.softmax = flags & 1; a.stats = scratch_dev;
    if (a.softmax && !scratch_dev) return fail(BO_E_ARG, "bo_nn_heads_f16: the softmax needs scratch_dev (20 * batch floats)");
    const dim3 pgrid((unsigned)((batch + 255) / 256), BO_HEADS_PS);
    if (a.softmax) {
        hipLaunchKernelGGL(bo_k_heads_policy_h<0>, pgrid, dim3(256), 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL(bo_k_heads_policy_h<1>, pgrid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(bo_k_heads_policy_h<2>, pgrid, dim3(256), 0, (hipStream_t)stream, a);
    }
    hipLaunchKernelGGL(bo_k_heads_value_h, dim3((unsigned)((batch + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

extern "C" int bo_nn_value_tail(const float *h_dev, const float *w_dev, const float *bias_dev, float *out_dev, int batch, int hidden, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)h_dev; (void)w_dev; (void)bias_dev; (void)out_dev; (void)batch; (void)hidden; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_value_tail is a gfx950-only kernel");
#else
    if (!h_dev || !w_dev || !bias_dev || !out_dev || batch < 1 || hidden < 1) return fail(BO_E_ARG, "bad arguments");
    hipLaunchKernelGGL(bo_k_value_tail, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h_dev, w_dev, bias_dev, out_dev, batch, hidden);
    RT((int)hipGetLastError());
    return BO_OK;
#endif
}

// small-batch form of bo_nn_conv3x3 (bo_conv.h: bo_k_conv3x3_small): (c_out/16) x 4 workgroups per board
extern "C" int bo_nn_conv3x3_small(const float *x_dev, const float *wpacked_dev, const float *bias_dev, const float *residual_dev, float *y_dev,
                                   int batch, int c_in, int c_in_x, int c_out, int mode, void *stream) {
#if defined(BO_WAVE_EMU)
    (void)x_dev; (void)wpacked_dev; (void)bias_dev; (void)residual_dev; (void)y_dev; (void)batch; (void)c_in; (void)c_in_x; (void)c_out; (void)mode; (void)stream;
    return fail(BO_E_CONFIG, "bo_nn_conv3x3_small is a gfx950-only kernel");
#else
    if (!x_dev || !wpacked_dev || !bias_dev || !y_dev || batch < 1 || batch > 65535) return fail(BO_E_ARG, "bad arguments");
    if (mode < 0 || mode > 2 || (mode == 2 && !residual_dev)) return fail(BO_E_ARG, "bad epilogue mode");
    if (c_in_x < 1 || c_in_x > c_in) return fail(BO_E_ARG, "c_in_x must be in 1..c_in");
    const bo_f32x4 *wp = reinterpret_cast<const bo_f32x4 *>(wpacked_dev);
    hipStream_t st = (hipStream_t)stream;
#define BO_SMALL_CASE(CI, CO)                                                                                                             \
    if (c_in == CI && c_out == CO) {                                                                                                      \
        hipLaunchKernelGGL((bo_k_conv3x3_small<CI, CO>), dim3(CO / 16, 4, (unsigned)batch), dim3(256), 0, st, x_dev, wp, bias_dev, residual_dev, \
                           y_dev, c_in_x, mode);                                                                                          \
        RT((int)hipGetLastError());                                                                                                       \
        return BO_OK;                                                                                                                     \
    }
    BO_SMALL_CASE(128, 64) BO_SMALL_CASE(64, 64) BO_SMALL_CASE(128, 128) BO_SMALL_CASE(128, 256) BO_SMALL_CASE(256, 256)
#undef BO_SMALL_CASE
    return fail(BO_E_CONFIG, "bo_nn_conv3x3_small: supported (c_in, c_out): (128 | C, C) for C in {64, 128, 256}");
#endif
}
