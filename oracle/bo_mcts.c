/*
 * oracle/bo_mcts.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Literal C restatement of /root/reference/mcts.py and the game loop of
 * /root/reference/self_play.py (see bo_oracle.h).  It deliberately keeps the
 * reference's control flow (one row per simulation, rows replayed one by one
 * in _evaluate_batch, children found by a per-row stable sort) so that the HIP
 * engine -- which uses an algebraically equivalent but differently organised
 * schedule -- is checked against the reference's own order of operations.
 *
 * Two deliberate, result-preserving shortcuts (stated in DESIGN.md):
 *  - is_terminal() of a node is cached (the reference recomputes it every
 *    simulation, mcts.py:152,235; it is a pure function of the node);
 *  - identical rows of one NN batch are evaluated once (regime R3: identical
 *    inputs give identical net outputs).
 */
#include "bo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* NumPy pairwise float32 sum (numpy/_core/src/umath/loops_utils.h.src,
 * @TYPE@_pairwise_sum, PW_BLOCKSIZE = 128, 8 accumulators).            */
/* ------------------------------------------------------------------ */
float bo_np_sum_f32(const float *a, long n) {
    if (n < 8) {
        float res = 0.0f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) {
            r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
            r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
        }
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return bo_np_sum_f32(a, n2) + bo_np_sum_f32(a + n2, n - n2);
    }
}

/* ------------------------------------------------------------------ */
/* tree                                                                */
/* ------------------------------------------------------------------ */
typedef struct {
    int parent;
    int *children; /* insertion order == dict order, mcts.py:33,70 */
    int n_children, cap_children;
    bo_move move;
    bo_pos pos;
    bo_key key;
    uint8_t irrev_in;
    int n_visits;
    float q_value;
    float prior;
    int terminal; /* -1 unknown, else bo_outcome_claim_draw */
} node_t;

typedef struct {
    node_t *nodes;
    int n, cap;
    const bo_oracle_config *cfg;
    const bo_oracle_callbacks *cb;
    const bo_stack *root_stack;
    bo_pos hist[8]; /* history BEFORE the node being encoded (<=7) */
    int n_hist;
    const bo_tracker *trk;
    bo_stack work;
    int err;
} tree_t;

static int new_node(tree_t *t, int parent, float prior, const bo_pos *pos, bo_move move, int irrev_in) {
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 256;
        t->nodes = (node_t *)realloc(t->nodes, sizeof(node_t) * (size_t)t->cap);
    }
    node_t *nd = &t->nodes[t->n];
    memset(nd, 0, sizeof(*nd));
    nd->parent = parent;
    nd->prior = prior;
    nd->pos = *pos;
    bo_key_of(pos, &nd->key); /* mcts.py:37 */
    nd->move = move;
    nd->irrev_in = (uint8_t)irrev_in;
    nd->n_visits = 0;
    nd->q_value = 0.0f;
    nd->terminal = -1;
    return t->n++;
}

static void add_child(node_t *nd, int c) {
    if (nd->n_children == nd->cap_children) {
        nd->cap_children = nd->cap_children ? nd->cap_children * 2 : 16;
        nd->children = (int *)realloc(nd->children, sizeof(int) * (size_t)nd->cap_children);
    }
    nd->children[nd->n_children++] = c;
}

/* MCTSNode.is_terminal, mcts.py:150-152: the node's board is a copy of the
 * root board (with the whole game's move stack) plus the path's moves. */
static int node_terminal(tree_t *t, int idx) {
    node_t *nd = &t->nodes[idx];
    if (nd->terminal >= 0) return nd->terminal;
    int path[4096], d = 0;
    for (int i = idx; i > 0; i = t->nodes[i].parent) path[d++] = i;
    bo_stack *w = &t->work;
    w->n = t->root_stack->n;
    for (int i = d - 1; i >= 0; i--) {
        node_t *p = &t->nodes[path[i]];
        bo_stack_reserve(w, w->n + 1);
        w->pos[w->n] = p->pos;
        w->key[w->n] = p->key;
        w->irrev_in[w->n] = p->irrev_in;
        w->n++;
    }
    nd->terminal = bo_outcome_claim_draw(w);
    w->n = t->root_stack->n;
    return nd->terminal;
}

/* MCTSNode.expand, mcts.py:45-70 */
static void expand(tree_t *t, int idx, const float *probs, const bo_move *legal, int n_legal) {
    node_t *nd = &t->nodes[idx];
    double w = t->cfg->widen_coeff * sqrt((double)nd->n_visits + 1.0);
    int max_children = (w != 0.0) ? (int)w : n_legal; /* int(x or len(legal)) */
    /* stable descending sort by prior (sorted(..., reverse=True) keeps ties in
     * python-chess order): insertion sort on an index array */
    int order[BO_MAX_MOVES];
    float key[BO_MAX_MOVES];
    for (int i = 0; i < n_legal; i++) {
        int mi = bo_move_to_index(legal[i]);
        key[i] = probs[mi];
        int j = i;
        while (j > 0 && key[order[j - 1]] < key[i]) { order[j] = order[j - 1]; j--; }
        order[j] = i;
    }
    int take = max_children < n_legal ? max_children : n_legal;
    if (take < 0) take = 0;
    for (int r = 0; r < take; r++) {
        bo_move m = legal[order[r]];
        int present = 0;
        nd = &t->nodes[idx];
        for (int c = 0; c < nd->n_children; c++) {
            bo_move cm = t->nodes[nd->children[c]].move;
            if (cm.from == m.from && cm.to == m.to && cm.promo == m.promo) { present = 1; break; }
        }
        if (present) continue;
        bo_pos child = nd->pos;
        int irrev = bo_is_irreversible(&nd->pos, m);
        bo_push(&child, m);
        int c = new_node(t, idx, key[order[r]], &child, m, irrev);
        add_child(&t->nodes[idx], c);
    }
}

/* MCTSNode.select_child, mcts.py:72-118 (regime R3: binary32, no FMA) */
static int select_child(tree_t *t, int idx) {
    node_t *nd = &t->nodes[idx];
    int parent_visits = nd->parent >= 0 ? t->nodes[nd->parent].n_visits : nd->n_visits; /* mcts.py:89 */
    float sqrt_parent = (float)sqrt((double)parent_visits + 1e-8);                      /* mcts.py:93 */
    float cpuct = (float)t->cfg->cpuct;
    float best = -INFINITY;
    int best_child = -1;
    for (int c = 0; c < nd->n_children; c++) {
        node_t *ch = &t->nodes[nd->children[c]];
        volatile float t1 = cpuct * ch->prior;
        volatile float t2 = t1 * sqrt_parent;
        float q, u;
        if (ch->n_visits > 0) {
            q = ch->q_value;
            u = t2 / (float)(1 + ch->n_visits);
        } else {
            q = 0.0f;
            u = t2;
        }
        float score = q + u;
        if (score > best) { best = score; best_child = nd->children[c]; }
    }
    if (best_child < 0) { /* all scores NaN: reference falls back to random.choice (mcts.py:110-116) */
        t->err = 1;
        best_child = nd->children[0];
    }
    return best_child;
}

/* MCTSNode.update / update_recursive, mcts.py:120-144 */
static void update_recursive(tree_t *t, int idx, float value) {
    while (idx >= 0) {
        node_t *nd = &t->nodes[idx];
        nd->n_visits += 1;
        volatile float d = value - nd->q_value;
        volatile float e = d / (float)nd->n_visits;
        nd->q_value = nd->q_value + e;
        value = -value;
        idx = nd->parent;
    }
}

static void encode_node(tree_t *t, int idx, float *planes) {
    bo_pos h[8];
    int n = 0;
    for (int i = 0; i < t->n_hist; i++) h[n++] = t->hist[i];
    h[n++] = t->nodes[idx].pos; /* (history + [board])[-8:], mcts.py:180,242 */
    bo_encode_board(h, n, t->trk, planes);
}

typedef struct {
    float *planes, *probs, *values;
    int cap;
} evalbuf_t;

static void evalbuf_reserve(evalbuf_t *b, int n) {
    if (n <= b->cap) return;
    b->cap = n;
    b->planes = (float *)realloc(b->planes, sizeof(float) * BO_PLANES_SIZE * (size_t)n);
    b->probs = (float *)realloc(b->probs, sizeof(float) * BO_NUM_ACTIONS * (size_t)n);
    b->values = (float *)realloc(b->values, sizeof(float) * (size_t)n);
}

/* _evaluate_batch, mcts.py:283-295 */
static int evaluate_batch(tree_t *t, const int *pending, int n_pending, evalbuf_t *eb, bo_oracle_result *out) {
    int uniq[4096], n_uniq = 0;
    int *slot = (int *)malloc(sizeof(int) * (size_t)n_pending);
    for (int r = 0; r < n_pending; r++) {
        int s = -1;
        for (int u = 0; u < n_uniq; u++)
            if (uniq[u] == pending[r]) { s = u; break; }
        if (s < 0) { s = n_uniq; uniq[n_uniq++] = pending[r]; }
        slot[r] = s;
    }
    evalbuf_reserve(eb, n_uniq);
    for (int u = 0; u < n_uniq; u++) encode_node(t, uniq[u], eb->planes + (size_t)u * BO_PLANES_SIZE);
    int rc = t->cb->eval(t->cb->user, eb->planes, n_uniq, eb->probs, eb->values);
    if (rc) { free(slot); return rc; }
    out->n_evals += n_uniq;
    out->n_batches += 1;
    out->n_batch_rows += n_pending;
    if (n_uniq > out->max_unique_in_batch) out->max_unique_in_batch = n_uniq;
    for (int r = 0; r < n_pending; r++) {
        int leaf = pending[r];
        bo_move legal[BO_MAX_MOVES];
        int n_legal = bo_legal_moves(&t->nodes[leaf].pos, legal);
        expand(t, leaf, eb->probs + (size_t)slot[r] * BO_NUM_ACTIONS, legal, n_legal);
        update_recursive(t, leaf, eb->values[slot[r]]);
    }
    free(slot);
    return 0;
}

void bo_oracle_result_free(bo_oracle_result *r) {
    free(r->nodes);
    r->nodes = NULL;
}

int bo_oracle_run_mcts(const bo_oracle_config *cfg, const bo_oracle_callbacks *cb, const bo_stack *board,
                       const bo_pos *hist, int n_hist, const bo_tracker *trk, bo_oracle_result *out) {
    bo_rules_init();
    memset(out, 0, sizeof(*out));
    tree_t t;
    memset(&t, 0, sizeof(t));
    t.cfg = cfg; t.cb = cb; t.root_stack = board; t.trk = trk;
    if (n_hist > 7) { hist += n_hist - 7; n_hist = 7; }
    for (int i = 0; i < n_hist; i++) t.hist[i] = hist[i];
    t.n_hist = n_hist;
    bo_stack_copy(&t.work, board);
    evalbuf_t eb = {0};
    int rc = 0;

    bo_move none = {0, 0, 0, 0};
    int root = new_node(&t, -1, 1.0f, bo_stack_top(board), none, board->irrev_in[board->n - 1]); /* mcts.py:176 */
    bo_move legal[BO_MAX_MOVES];
    int n_legal = bo_legal_moves(&t.nodes[root].pos, legal);

    if (!node_terminal(&t, root)) { /* mcts.py:179-203 */
        evalbuf_reserve(&eb, 1);
        encode_node(&t, root, eb.planes);
        rc = cb->eval(cb->user, eb.planes, 1, eb.probs, eb.values);
        if (rc) goto done;
        out->n_evals += 1;
        float *p = eb.probs;
        expand(&t, root, p, legal, n_legal); /* mcts.py:186 */
        if (cfg->dirichlet_alpha > 0) {      /* mcts.py:190-201 */
            double noise[BO_MAX_MOVES];
            rc = cb->noise(cb->user, n_legal, noise);
            if (rc) goto done;
            float keep = (float)(1.0 - cfg->dirichlet_eps);
            for (int i = 0; i < n_legal; i++) {
                int idx = bo_move_to_index(legal[i]);
                volatile float a = keep * p[idx];
                double s = (double)a + cfg->dirichlet_eps * noise[i];
                p[idx] = (float)s;
            }
            volatile float sum = bo_np_sum_f32(p, BO_NUM_ACTIONS);
            volatile float denom = sum + (float)1e-12;
            for (int i = 0; i < BO_NUM_ACTIONS; i++) p[i] = p[i] / denom;
        }
        expand(&t, root, p, legal, n_legal); /* mcts.py:203 */
    }

    {
        int *pending = (int *)malloc(sizeof(int) * (size_t)(cfg->batch_size > 0 ? cfg->batch_size : 1));
        int n_pending = 0;
        for (int sim = 0; sim < cfg->num_simulations; sim++) { /* mcts.py:210 */
            int node = root;
            while (t.nodes[node].n_children > 0) node = select_child(&t, node);
            int term = node_terminal(&t, node);
            if (term) { /* mcts.py:235-238 */
                update_recursive(&t, node, term == BO_CHECKMATE ? 1.0f : 0.0f);
                out->n_terminal_sims++;
                continue;
            }
            pending[n_pending++] = node;
            if (n_pending >= cfg->batch_size) { /* mcts.py:251-254 */
                rc = evaluate_batch(&t, pending, n_pending, &eb, out);
                n_pending = 0;
                if (rc) break;
            }
        }
        if (!rc && n_pending) rc = evaluate_batch(&t, pending, n_pending, &eb, out); /* mcts.py:256-257 */
        free(pending);
        if (rc) goto done;
    }

    /* mcts.py:259-280 */
    memset(out->pi, 0, sizeof(out->pi));
    if (n_legal == 0) {
        out->status = 1;
    } else {
        long total = 0;
        int visits[BO_MAX_MOVES];
        for (int i = 0; i < n_legal; i++) {
            visits[i] = 0;
            node_t *r = &t.nodes[root];
            for (int c = 0; c < r->n_children; c++) {
                bo_move cm = t.nodes[r->children[c]].move;
                if (cm.from == legal[i].from && cm.to == legal[i].to && cm.promo == legal[i].promo)
                    visits[i] = t.nodes[r->children[c]].n_visits;
            }
            total += visits[i];
        }
        int best = 0;
        for (int i = 0; i < n_legal; i++) {
            int idx = bo_move_to_index(legal[i]);
            if (total > 0) out->pi[idx] = (float)((double)visits[i] / (double)total);
            else out->pi[idx] = (float)(1.0 / (double)n_legal);
            if (visits[i] > visits[best]) best = i;
        }
        out->best_move = legal[best];
    }

    out->n_nodes = t.n;
    out->nodes = (bo_oracle_node *)malloc(sizeof(bo_oracle_node) * (size_t)t.n);
    for (int i = 0; i < t.n; i++) {
        out->nodes[i].parent = t.nodes[i].parent;
        out->nodes[i].n_visits = t.nodes[i].n_visits;
        out->nodes[i].q_value = t.nodes[i].q_value;
        out->nodes[i].prior = t.nodes[i].prior;
        out->nodes[i].move = t.nodes[i].move;
        out->nodes[i].n_children = t.nodes[i].n_children;
        out->nodes[i].terminal = t.nodes[i].terminal;
    }
done:
    if (t.err && !rc) rc = -100; /* NaN scores: reference behaviour is random */
    for (int i = 0; i < t.n; i++) free(t.nodes[i].children);
    free(t.nodes);
    bo_stack_free(&t.work);
    free(eb.planes); free(eb.probs); free(eb.values);
    return rc;
}

/* ------------------------------------------------------------------ */
/* run_self_play_game, self_play.py:84-216                             */
/* ------------------------------------------------------------------ */
void bo_game_result_free(bo_game_result *r) {
    free(r->records); free(r->moves);
    r->records = NULL; r->moves = NULL;
}

int bo_oracle_self_play(const bo_oracle_config *cfg, const bo_oracle_callbacks *cb, const char *start_fen,
                        int max_plies, bo_game_result *out) {
    bo_rules_init();
    memset(out, 0, sizeof(*out));
    bo_pos start;
    if (start_fen && *start_fen) {
        if (bo_pos_from_fen(start_fen, &start)) return -1;
    } else bo_pos_startpos(&start);

    bo_stack board;
    bo_stack_init(&board, &start);
    bo_tracker trk = {0};
    bo_tracker_add(&trk, &board.key[0]); /* self_play.py:93 */

    int cap = 256, n_states = 0;
    float *pis = (float *)malloc(sizeof(float) * BO_NUM_ACTIONS * (size_t)cap);
    bo_move *moves = (bo_move *)malloc(sizeof(bo_move) * (size_t)cap);
    int rc = 0, move_count = 0;

    while (bo_termination_claim_draw(&board) == 0 && move_count < cfg->max_game_moves &&
           (max_plies <= 0 || move_count < max_plies)) { /* self_play.py:101-103 */
        const bo_pos *cur = bo_stack_top(&board);
        int move_number = cur->fullmove_number; /* self_play.py:104 */
        /* board_history[max(0,len-8):-1], self_play.py:109 (board_history[i] == board.pos[i]) */
        int len = board.n;
        int h0 = len - 8 > 0 ? len - 8 : 0;
        bo_oracle_result res;
        rc = bo_oracle_run_mcts(cfg, cb, &board, &board.pos[h0], len - 1 - h0, &trk, &res);
        if (rc) { bo_oracle_result_free(&res); break; }
        if (res.status) { bo_oracle_result_free(&res); rc = -2; break; }
        out->n_sims += cfg->num_simulations;
        out->n_evals += res.n_evals;
        if (n_states == cap) {
            cap *= 2;
            pis = (float *)realloc(pis, sizeof(float) * BO_NUM_ACTIONS * (size_t)cap);
            moves = (bo_move *)realloc(moves, sizeof(bo_move) * (size_t)cap);
        }
        memcpy(pis + (size_t)n_states * BO_NUM_ACTIONS, res.pi, sizeof(res.pi)); /* self_play.py:122 */

        int action = cb->choose(cb->user, res.pi, move_number); /* self_play.py:125 */
        bo_move played;
        if (bo_index_to_move(action, cur, &played) != 0) played = res.best_move; /* self_play.py:127-137 */
        bo_move legal[BO_MAX_MOVES];
        int n_legal = bo_legal_moves(cur, legal), ok = 0, best_ok = 0;
        for (int i = 0; i < n_legal; i++) {
            if (legal[i].from == played.from && legal[i].to == played.to && legal[i].promo == played.promo) ok = 1;
            if (legal[i].from == res.best_move.from && legal[i].to == res.best_move.to &&
                legal[i].promo == res.best_move.promo) best_ok = 1;
        }
        if (!ok) { /* self_play.py:142-167 */
            int same = played.from == res.best_move.from && played.to == res.best_move.to &&
                       played.promo == res.best_move.promo;
            if (!same && best_ok) played = res.best_move;
            else { bo_oracle_result_free(&res); rc = -3; break; }
        }
        bo_oracle_result_free(&res);
        moves[n_states] = played;
        n_states++;
        bo_stack_push(&board, played);                 /* self_play.py:171 */
        bo_tracker_add(&trk, &board.key[board.n - 1]); /* self_play.py:182 */
        move_count++;
    }

    if (!rc) {
        int term = bo_termination_claim_draw(&board);
        float outcome = term == 1 ? 1.0f : 0.0f; /* utils.get_game_outcome, utils.py:385-396 */
        out->termination = term;
        out->outcome = outcome;
        out->n_records = n_states;
        out->n_moves = n_states;
        out->records = (bo_record *)malloc(sizeof(bo_record) * (size_t)(n_states ? n_states : 1));
        out->moves = moves;
        moves = NULL;
        for (int i = 0; i < n_states; i++) { /* self_play.py:200-208 */
            const bo_pos *st = &board.pos[i];
            out->records[i].z = st->turn == BO_WHITE ? outcome : -outcome;
            int h0 = i + 1 - 8 > 0 ? i + 1 - 8 : 0;
            bo_encode_board(&board.pos[h0], i + 1 - h0, &trk, out->records[i].state);
            memcpy(out->records[i].pi, pis + (size_t)i * BO_NUM_ACTIONS, sizeof(float) * BO_NUM_ACTIONS);
        }
    }
    out->status = rc;
    free(pis); free(moves);
    bo_tracker_free(&trk);
    bo_stack_free(&board);
    return rc;
}
