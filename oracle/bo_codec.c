/*
 * oracle/bo_codec.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Restatement of the board / move codec of the reference:
 *   move_to_index   /root/reference/utils.py:221-281
 *   index_to_move   /root/reference/utils.py:284-365
 *   encode_board    /root/reference/utils.py:111-217
 *   RepetitionTracker.repetitions  /root/reference/utils.py:91-99
 * Constants: PIECE_ORDER utils.py:15-28, QUEEN/KNIGHT/PROMOTION directions
 * utils.py:34-62, INPUT_CHANNELS/NUM_ACTIONS config.py:28-29.
 */
#include "bo_oracle.h"

#include <stdlib.h>
#include <string.h>

#define RANK_OF(sq) ((sq) >> 3)
#define FILE_OF(sq) ((sq) & 7)

/* (d_rank, d_file): N, NE, E, SE, S, SW, W, NW  -- utils.py:34-43 */
static const int QDIR[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
/* utils.py:45-54 */
static const int NDIR[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
/* (d_file, d_rank) for White -- utils.py:56-60 */
static const int PDIR[3][2] = {{-1, 1}, {0, 1}, {1, 1}};

int bo_move_to_index(bo_move m) {
    int fr = RANK_OF(m.from), ff = FILE_OF(m.from);
    int tr = RANK_OF(m.to), tf = FILE_OF(m.to);
    int dr = tr - fr, df = tf - ff;
    if (m.promo && m.promo != BO_QUEEN) { /* utils.py:235-248 */
        int d_r;
        if (fr == 6) d_r = dr;
        else if (fr == 1) d_r = -dr;
        else return -1;
        int dir = -1;
        for (int i = 0; i < 3; i++)
            if (PDIR[i][0] == df && PDIR[i][1] == d_r) dir = i;
        if (dir < 0) return -1;
        int piece = m.promo == BO_KNIGHT ? 0 : m.promo == BO_BISHOP ? 1 : m.promo == BO_ROOK ? 2 : -1;
        if (piece < 0) return -1;
        return m.from * 73 + 64 + piece * 3 + dir;
    }
    int adr = abs(dr), adf = abs(df);
    if ((adr == 1 && adf == 2) || (adr == 2 && adf == 1)) { /* utils.py:251-260 */
        for (int i = 0; i < 8; i++)
            if (NDIR[i][0] == dr && NDIR[i][1] == df) return m.from * 73 + 56 + i;
        return -1;
    }
    if (adr == adf || dr == 0 || df == 0) { /* utils.py:263-279 */
        int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
        int dir = -1;
        for (int i = 0; i < 8; i++)
            if (QDIR[i][0] == sr && QDIR[i][1] == sf) dir = i;
        int dist = adr > adf ? adr : adf;
        if (dir < 0 || dist == 0 || dist > 7) return -1;
        return m.from * 73 + dir * 7 + (dist - 1);
    }
    return -1;
}

int bo_index_to_move(int index, const bo_pos *board, bo_move *out) {
    if (index < 0 || index >= BO_NUM_ACTIONS) return -1;
    int from = index / 73, plane = index % 73;
    int fr = RANK_OF(from), ff = FILE_OF(from);
    out->from = (uint8_t)from; out->promo = 0; out->pad = 0;
    if (plane < 56) { /* utils.py:300-319 */
        int dir = plane / 7, dist = plane % 7 + 1;
        int tr = fr + QDIR[dir][0] * dist, tf = ff + QDIR[dir][1] * dist;
        if (tr < 0 || tr > 7 || tf < 0 || tf > 7) return -2;
        out->to = (uint8_t)(tr * 8 + tf);
        if (bo_piece_type_at(board, from) == BO_PAWN) {
            int c = bo_color_at(board, from);
            if ((c == BO_WHITE && fr == 6 && tr == 7) || (c == BO_BLACK && fr == 1 && tr == 0))
                out->promo = BO_QUEEN;
        }
        return 0;
    }
    if (plane < 64) { /* utils.py:322-332 */
        int d = plane - 56;
        int tr = fr + NDIR[d][0], tf = ff + NDIR[d][1];
        if (tr < 0 || tr > 7 || tf < 0 || tf > 7) return -2;
        out->to = (uint8_t)(tr * 8 + tf);
        return 0;
    }
    /* utils.py:335-365 */
    int off = plane - 64, piece = off / 3, dir = off % 3;
    static const int PROMO[3] = {BO_KNIGHT, BO_BISHOP, BO_ROOK};
    if (bo_piece_type_at(board, from) != BO_PAWN) return -3;
    int c = bo_color_at(board, from);
    int df = PDIR[dir][0], dr;
    if (c == BO_WHITE && fr == 6) dr = PDIR[dir][1];
    else if (c == BO_BLACK && fr == 1) dr = -PDIR[dir][1];
    else return -3;
    int tr = fr + dr, tf = ff + df;
    if (tr < 0 || tr > 7 || tf < 0 || tf > 7) return -2;
    out->to = (uint8_t)(tr * 8 + tf);
    out->promo = (uint8_t)PROMO[piece];
    return 0;
}

int bo_tracker_count(const bo_tracker *t, const bo_key *k) {
    for (int i = 0; i < t->n; i++)
        if (bo_key_eq(&t->keys[i], k)) return t->counts[i];
    return 0;
}

void bo_tracker_add(bo_tracker *t, const bo_key *k) {
    for (int i = 0; i < t->n; i++)
        if (bo_key_eq(&t->keys[i], k)) { t->counts[i]++; return; }
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->keys = (bo_key *)realloc(t->keys, sizeof(bo_key) * (size_t)t->cap);
        t->counts = (int *)realloc(t->counts, sizeof(int) * (size_t)t->cap);
    }
    t->keys[t->n] = *k;
    t->counts[t->n] = 1;
    t->n++;
}

void bo_tracker_free(bo_tracker *t) {
    free(t->keys); free(t->counts);
    memset(t, 0, sizeof(*t));
}

static void fill_plane(float *planes, int plane, float v) {
    for (int i = 0; i < 64; i++) planes[plane * 64 + i] = v;
}

static void piece_plane(float *planes, int plane, bb_t bbmask) {
    while (bbmask) {
        int sq = __builtin_ctzll(bbmask);
        bbmask &= bbmask - 1;
        planes[plane * 64 + sq] = 1.0f;
    }
}

/* encode_board(board, history, tracker); history[n-1] must be the board. */
void bo_encode_board(const bo_pos *hist, int n_hist, const bo_tracker *trk, float *planes) {
    memset(planes, 0, sizeof(float) * BO_INPUT_CHANNELS * 64);
    if (n_hist > 8) { hist += n_hist - 8; n_hist = 8; } /* utils.py:140-142 */
    const bo_pos *board = &hist[n_hist - 1];
    int start = (8 - n_hist) * 14; /* utils.py:163 */
    for (int i = 0; i < n_hist; i++) {
        const bo_pos *h = &hist[i];
        int base = start + i * 14;
        /* PIECE_ORDER utils.py:15-28: P,p,N,n,B,b,R,r,Q,q,K,k */
        const bb_t types[6] = {h->pawns, h->knights, h->bishops, h->rooks, h->queens, h->kings};
        for (int t = 0; t < 6; t++) {
            piece_plane(planes, base + 2 * t, types[t] & h->occ[BO_WHITE]);
            piece_plane(planes, base + 2 * t + 1, types[t] & h->occ[BO_BLACK]);
        }
        bo_key k;
        bo_key_of(h, &k);
        int rep = bo_tracker_count(trk, &k) - 1; /* utils.py:91-99 */
        if (rep < 0) rep = 0;
        fill_plane(planes, base + 12, rep >= 1 ? 1.0f : 0.0f);
        fill_plane(planes, base + 13, rep >= 2 ? 1.0f : 0.0f);
    }
    fill_plane(planes, 112, board->turn == BO_WHITE ? 1.0f : 0.0f);
    fill_plane(planes, 113, (board->castling & ((bb_t)1 << 7)) ? 1.0f : 0.0f);
    fill_plane(planes, 114, (board->castling & ((bb_t)1 << 0)) ? 1.0f : 0.0f);
    fill_plane(planes, 115, (board->castling & ((bb_t)1 << 63)) ? 1.0f : 0.0f);
    fill_plane(planes, 116, (board->castling & ((bb_t)1 << 56)) ? 1.0f : 0.0f);
    fill_plane(planes, 117, (float)board->halfmove_clock);
    fill_plane(planes, 118, (float)board->fullmove_number);
    if (board->ep_square >= 0) planes[119 * 64 + board->ep_square] = 1.0f;
}
