/*
 * oracle/bo_rules.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See bo_rules.h for what is restated and the parity status.
 */
#include "bo_rules.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define BB(sq) ((bb_t)1 << (sq))
#define RANK_OF(sq) ((sq) >> 3)
#define FILE_OF(sq) ((sq) & 7)

static const bb_t BB_RANK_1 = 0xffULL;
static const bb_t BB_RANK_8 = 0xffULL << 56;
static const bb_t BB_RANK_3 = 0xffULL << 16;
static const bb_t BB_RANK_4 = 0xffULL << 24;
static const bb_t BB_RANK_5 = 0xffULL << 32;
static const bb_t BB_RANK_6 = 0xffULL << 40;
static const bb_t BB_DARK = 0xaa55aa55aa55aa55ULL;
static const bb_t BB_LIGHT = 0x55aa55aa55aa55aaULL;

static bb_t KNIGHT_ATT[64], KING_ATT[64], PAWN_ATT[2][64];
static bb_t BETWEEN[64][64], RAY[64][64];
static int g_init = 0;

static inline int msb(bb_t b) { return 63 - __builtin_clzll(b); }
static inline int popcnt(bb_t b) { return __builtin_popcountll(b); }

static bb_t step_att(int sq, const int (*d)[2], int nd) {
    bb_t a = 0;
    for (int i = 0; i < nd; i++) {
        int r = RANK_OF(sq) + d[i][0], f = FILE_OF(sq) + d[i][1];
        if (r >= 0 && r < 8 && f >= 0 && f < 8) a |= BB(r * 8 + f);
    }
    return a;
}

static bb_t slide_att(int sq, bb_t occ, const int (*d)[2], int nd) {
    bb_t a = 0;
    for (int i = 0; i < nd; i++) {
        int r = RANK_OF(sq) + d[i][0], f = FILE_OF(sq) + d[i][1];
        while (r >= 0 && r < 8 && f >= 0 && f < 8) {
            bb_t b = BB(r * 8 + f);
            a |= b;
            if (occ & b) break;
            r += d[i][0];
            f += d[i][1];
        }
    }
    return a;
}

static const int ROOK_D[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
static const int BISH_D[4][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}};

static inline bb_t rook_att(int sq, bb_t occ) { return slide_att(sq, occ, ROOK_D, 4); }
static inline bb_t bishop_att(int sq, bb_t occ) { return slide_att(sq, occ, BISH_D, 4); }

void bo_rules_init(void) {
    if (g_init) return;
    static const int KN[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
    static const int KG[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    static const int PW[2][2] = {{1, -1}, {1, 1}};
    static const int PB[2][2] = {{-1, -1}, {-1, 1}};
    for (int s = 0; s < 64; s++) {
        KNIGHT_ATT[s] = step_att(s, KN, 8);
        KING_ATT[s] = step_att(s, KG, 8);
        PAWN_ATT[BO_WHITE][s] = step_att(s, PW, 2);
        PAWN_ATT[BO_BLACK][s] = step_att(s, PB, 2);
    }
    for (int a = 0; a < 64; a++) {
        for (int b = 0; b < 64; b++) {
            BETWEEN[a][b] = 0;
            RAY[a][b] = 0;
            if (a == b) continue;
            int dr = RANK_OF(b) - RANK_OF(a), df = FILE_OF(b) - FILE_OF(a);
            if (!(dr == 0 || df == 0 || abs(dr) == abs(df))) continue;
            int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
            /* squares strictly between */
            int r = RANK_OF(a) + sr, f = FILE_OF(a) + sf;
            while (r * 8 + f != b) {
                BETWEEN[a][b] |= BB(r * 8 + f);
                r += sr;
                f += sf;
            }
            /* whole line through a and b, edge to edge */
            bb_t line = BB(a);
            r = RANK_OF(a) + sr; f = FILE_OF(a) + sf;
            while (r >= 0 && r < 8 && f >= 0 && f < 8) { line |= BB(r * 8 + f); r += sr; f += sf; }
            r = RANK_OF(a) - sr; f = FILE_OF(a) - sf;
            while (r >= 0 && r < 8 && f >= 0 && f < 8) { line |= BB(r * 8 + f); r -= sr; f -= sf; }
            RAY[a][b] = line;
        }
    }
    g_init = 1;
}

/* ------------------------------------------------------------------ */
/* position helpers                                                    */
/* ------------------------------------------------------------------ */

static inline bb_t occ_all(const bo_pos *p) { return p->occ[0] | p->occ[1]; }

int bo_piece_type_at(const bo_pos *p, int sq) {
    bb_t b = BB(sq);
    if (!(occ_all(p) & b)) return 0;
    if (p->pawns & b) return BO_PAWN;
    if (p->knights & b) return BO_KNIGHT;
    if (p->bishops & b) return BO_BISHOP;
    if (p->rooks & b) return BO_ROOK;
    if (p->queens & b) return BO_QUEEN;
    return BO_KING;
}

int bo_color_at(const bo_pos *p, int sq) { return (p->occ[BO_WHITE] & BB(sq)) ? BO_WHITE : BO_BLACK; }

static void remove_piece(bo_pos *p, int sq) {
    bb_t m = ~BB(sq);
    p->pawns &= m; p->knights &= m; p->bishops &= m;
    p->rooks &= m; p->queens &= m; p->kings &= m;
    p->occ[0] &= m; p->occ[1] &= m;
}

static void set_piece(bo_pos *p, int sq, int pt, int color) {
    remove_piece(p, sq);
    bb_t b = BB(sq);
    switch (pt) {
    case BO_PAWN: p->pawns |= b; break;
    case BO_KNIGHT: p->knights |= b; break;
    case BO_BISHOP: p->bishops |= b; break;
    case BO_ROOK: p->rooks |= b; break;
    case BO_QUEEN: p->queens |= b; break;
    default: p->kings |= b; break;
    }
    p->occ[color] |= b;
}

/* python-chess Board.clean_castling_rights() for standard chess */
static bb_t clean_castling(const bo_pos *p, bb_t rights) {
    bb_t c = rights & p->rooks;
    bb_t w = c & BB_RANK_1 & p->occ[BO_WHITE] & (BB(0) | BB(7));
    bb_t b = c & BB_RANK_8 & p->occ[BO_BLACK] & (BB(56) | BB(63));
    if (!(p->occ[BO_WHITE] & p->kings & BB(4))) w = 0;
    if (!(p->occ[BO_BLACK] & p->kings & BB(60))) b = 0;
    return w | b;
}

void bo_pos_startpos(bo_pos *out) {
    bo_pos_from_fen("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", out);
}

int bo_pos_from_fen(const char *fen, bo_pos *out) {
    bo_rules_init();
    memset(out, 0, sizeof(*out));
    out->turn = BO_WHITE;
    out->ep_square = -1;
    out->halfmove_clock = 0;
    out->fullmove_number = 1;
    const char *s = fen;
    while (*s == ' ') s++;
    int r = 7, f = 0;
    for (; *s && *s != ' '; s++) {
        char c = *s;
        if (c == '/') { r--; f = 0; continue; }
        if (c >= '1' && c <= '8') { f += c - '0'; continue; }
        int color = isupper((unsigned char)c) ? BO_WHITE : BO_BLACK;
        int pt;
        switch (tolower((unsigned char)c)) {
        case 'p': pt = BO_PAWN; break;
        case 'n': pt = BO_KNIGHT; break;
        case 'b': pt = BO_BISHOP; break;
        case 'r': pt = BO_ROOK; break;
        case 'q': pt = BO_QUEEN; break;
        case 'k': pt = BO_KING; break;
        default: return -1;
        }
        if (r < 0 || f > 7) return -1;
        set_piece(out, r * 8 + f, pt, color);
        f++;
    }
    while (*s == ' ') s++;
    if (*s) {
        if (*s == 'w') out->turn = BO_WHITE;
        else if (*s == 'b') out->turn = BO_BLACK;
        else return -1;
        s++;
    }
    while (*s == ' ') s++;
    bb_t rights = 0;
    if (*s) {
        for (; *s && *s != ' '; s++) {
            switch (*s) {
            case 'K': rights |= BB(7); break;
            case 'Q': rights |= BB(0); break;
            case 'k': rights |= BB(63); break;
            case 'q': rights |= BB(56); break;
            case '-': break;
            default: return -1;
            }
        }
    }
    out->castling = clean_castling(out, rights);
    while (*s == ' ') s++;
    if (*s) {
        if (*s != '-') {
            if (s[0] < 'a' || s[0] > 'h' || s[1] < '1' || s[1] > '8') return -1;
            out->ep_square = (s[1] - '1') * 8 + (s[0] - 'a');
            s += 2;
        } else s++;
    }
    while (*s == ' ') s++;
    if (*s) { out->halfmove_clock = (int)strtol(s, (char **)&s, 10); }
    while (*s == ' ') s++;
    if (*s) {
        out->fullmove_number = (int)strtol(s, (char **)&s, 10);
        if (out->fullmove_number < 1) out->fullmove_number = 1;
    }
    return 0;
}

void bo_pos_to_fen(const bo_pos *p, char *buf, int buflen) {
    char tmp[128];
    int n = 0;
    for (int r = 7; r >= 0; r--) {
        int empty = 0;
        for (int f = 0; f < 8; f++) {
            int sq = r * 8 + f, pt = bo_piece_type_at(p, sq);
            if (!pt) { empty++; continue; }
            if (empty) { tmp[n++] = (char)('0' + empty); empty = 0; }
            char c = " pnbrqk"[pt];
            if (bo_color_at(p, sq) == BO_WHITE) c = (char)toupper((unsigned char)c);
            tmp[n++] = c;
        }
        if (empty) tmp[n++] = (char)('0' + empty);
        if (r) tmp[n++] = '/';
    }
    tmp[n] = 0;
    char cast[8];
    int c = 0;
    if (p->castling & BB(7)) cast[c++] = 'K';
    if (p->castling & BB(0)) cast[c++] = 'Q';
    if (p->castling & BB(63)) cast[c++] = 'k';
    if (p->castling & BB(56)) cast[c++] = 'q';
    if (!c) cast[c++] = '-';
    cast[c] = 0;
    char ep[4] = "-";
    if (p->ep_square >= 0 && bo_has_legal_en_passant(p)) {
        ep[0] = (char)('a' + FILE_OF(p->ep_square));
        ep[1] = (char)('1' + RANK_OF(p->ep_square));
        ep[2] = 0;
    }
    snprintf(buf, (size_t)buflen, "%s %c %s %s %d %d", tmp, p->turn == BO_WHITE ? 'w' : 'b', cast, ep,
             p->halfmove_clock, p->fullmove_number);
}

/* ------------------------------------------------------------------ */
/* attacks                                                             */
/* ------------------------------------------------------------------ */

/* python-chess Board._attackers_mask(color, square, occupied) */
static bb_t attackers_mask(const bo_pos *p, int color, int sq, bb_t occ) {
    bb_t qr = p->queens | p->rooks, qb = p->queens | p->bishops;
    bb_t a = (KING_ATT[sq] & p->kings) | (KNIGHT_ATT[sq] & p->knights) | (rook_att(sq, occ) & qr) |
             (bishop_att(sq, occ) & qb) | (PAWN_ATT[!color][sq] & p->pawns);
    return a & p->occ[color];
}

/* python-chess Board.attacks_mask(square) for a non-pawn piece */
static bb_t piece_attacks(const bo_pos *p, int sq) {
    bb_t b = BB(sq), occ = occ_all(p);
    if (p->knights & b) return KNIGHT_ATT[sq];
    if (p->kings & b) return KING_ATT[sq];
    bb_t a = 0;
    if ((p->bishops | p->queens) & b) a |= bishop_att(sq, occ);
    if ((p->rooks | p->queens) & b) a |= rook_att(sq, occ);
    return a;
}

static int attacked_for_king(const bo_pos *p, bb_t path, bb_t occ) {
    while (path) {
        int sq = msb(path);
        path ^= BB(sq);
        if (attackers_mask(p, !p->turn, sq, occ)) return 1;
    }
    return 0;
}

int bo_is_check(const bo_pos *p) {
    bb_t k = p->kings & p->occ[p->turn];
    if (!k) return 0;
    return attackers_mask(p, !p->turn, msb(k), occ_all(p)) != 0;
}

/* ------------------------------------------------------------------ */
/* move generation, python-chess order                                 */
/* ------------------------------------------------------------------ */

typedef struct { bo_move *m; int n; } mlist;

static inline void emit(mlist *l, int from, int to, int promo) {
    bo_move mv; mv.from = (uint8_t)from; mv.to = (uint8_t)to; mv.promo = (uint8_t)promo; mv.pad = 0;
    l->m[l->n++] = mv;
}

static void emit_pawn(mlist *l, int from, int to) {
    int r = RANK_OF(to);
    if (r == 0 || r == 7) {
        emit(l, from, to, BO_QUEEN); emit(l, from, to, BO_ROOK);
        emit(l, from, to, BO_BISHOP); emit(l, from, to, BO_KNIGHT);
    } else emit(l, from, to, 0);
}

/* Board.generate_pseudo_legal_ep */
static void gen_pseudo_ep(const bo_pos *p, bb_t from_mask, bb_t to_mask, mlist *l) {
    if (p->ep_square < 0 || !(BB(p->ep_square) & to_mask)) return;
    if (BB(p->ep_square) & occ_all(p)) return;
    bb_t rank = p->turn == BO_WHITE ? BB_RANK_5 : BB_RANK_4;
    bb_t cap = p->pawns & p->occ[p->turn] & from_mask & PAWN_ATT[!p->turn][p->ep_square] & rank;
    while (cap) {
        int c = msb(cap);
        cap ^= BB(c);
        emit(l, c, p->ep_square, 0);
    }
}

/* Board.generate_castling_moves (standard chess) */
static void gen_castling(const bo_pos *p, bb_t from_mask, bb_t to_mask, mlist *l) {
    bb_t backrank = p->turn == BO_WHITE ? BB_RANK_1 : BB_RANK_8;
    bb_t king = p->occ[p->turn] & p->kings & backrank & from_mask;
    king &= (~king + 1);
    if (!king) return;
    bb_t all = occ_all(p);
    int ksq = msb(king);
    bb_t cand = p->castling & backrank & to_mask;
    while (cand) {
        int c = msb(cand);
        cand ^= BB(c);
        bb_t rook = BB(c);
        int a_side = rook < king;
        int base = p->turn == BO_WHITE ? 0 : 56;
        int kto = base + (a_side ? 2 : 6), rto = base + (a_side ? 3 : 5);
        bb_t king_to = BB(kto), rook_to = BB(rto);
        bb_t king_path = BETWEEN[ksq][kto], rook_path = BETWEEN[c][rto];
        if ((all ^ king ^ rook) & (king_path | rook_path | king_to | rook_to)) continue;
        if (attacked_for_king(p, king_path | king, all ^ king)) continue;
        if (attacked_for_king(p, king_to, all ^ king ^ rook ^ rook_to)) continue;
        emit(l, ksq, kto, 0);
    }
}

/* Board.generate_pseudo_legal_moves(from_mask, to_mask) */
static void gen_pseudo(const bo_pos *p, bb_t from_mask, bb_t to_mask, mlist *l) {
    bb_t our = p->occ[p->turn], all = occ_all(p);
    bb_t non_pawns = our & ~p->pawns & from_mask;
    while (non_pawns) {
        int from = msb(non_pawns);
        non_pawns ^= BB(from);
        bb_t mv = piece_attacks(p, from) & ~our & to_mask;
        while (mv) {
            int to = msb(mv);
            mv ^= BB(to);
            emit(l, from, to, 0);
        }
    }
    if (from_mask & p->kings) gen_castling(p, from_mask, to_mask, l);

    bb_t pawns = p->pawns & our & from_mask;
    if (!pawns) return;

    bb_t capturers = pawns;
    while (capturers) {
        int from = msb(capturers);
        capturers ^= BB(from);
        bb_t t = PAWN_ATT[p->turn][from] & p->occ[!p->turn] & to_mask;
        while (t) {
            int to = msb(t);
            t ^= BB(to);
            emit_pawn(l, from, to);
        }
    }
    bb_t single, dbl;
    if (p->turn == BO_WHITE) {
        single = (pawns << 8) & ~all;
        dbl = (single << 8) & ~all & (BB_RANK_3 | BB_RANK_4);
    } else {
        single = (pawns >> 8) & ~all;
        dbl = (single >> 8) & ~all & (BB_RANK_6 | BB_RANK_5);
    }
    single &= to_mask;
    dbl &= to_mask;
    while (single) {
        int to = msb(single);
        single ^= BB(to);
        emit_pawn(l, to + (p->turn == BO_BLACK ? 8 : -8), to);
    }
    while (dbl) {
        int to = msb(dbl);
        dbl ^= BB(to);
        emit(l, to + (p->turn == BO_BLACK ? 16 : -16), to, 0);
    }
    if (p->ep_square >= 0) gen_pseudo_ep(p, from_mask, to_mask, l);
}

/* Board._slider_blockers(king) */
static bb_t slider_blockers(const bo_pos *p, int king) {
    bb_t rq = p->rooks | p->queens, bq = p->bishops | p->queens;
    bb_t snipers = (rook_att(king, 0) & rq) | (bishop_att(king, 0) & bq);
    bb_t blockers = 0, all = occ_all(p);
    snipers &= p->occ[!p->turn];
    while (snipers) {
        int s = msb(snipers);
        snipers ^= BB(s);
        bb_t b = BETWEEN[king][s] & all;
        if (b && (b & (b - 1)) == 0) blockers |= b;
    }
    return blockers & p->occ[p->turn];
}

static int is_en_passant(const bo_pos *p, bo_move m) {
    int d = (int)m.to - (int)m.from;
    if (d < 0) d = -d;
    return p->ep_square == (int)m.to && (p->pawns & BB(m.from)) && (d == 7 || d == 9) &&
           !(occ_all(p) & BB(m.to));
}

static int is_castling_move(const bo_pos *p, bo_move m) {
    if (!(p->kings & BB(m.from))) return 0;
    int d = FILE_OF(m.from) - FILE_OF(m.to);
    if (d < 0) d = -d;
    return d > 1 || ((p->rooks & p->occ[p->turn] & BB(m.to)) != 0);
}

/* legality of an ep capture: play it on scratch bitboards and test the king
 * (same truth value as python-chess's pin_mask + _ep_skewered test). */
static int ep_is_safe(const bo_pos *p, int king, bo_move m) {
    bo_pos t = *p;
    int capsq = p->ep_square + (p->turn == BO_WHITE ? -8 : 8);
    remove_piece(&t, capsq);
    remove_piece(&t, m.from);
    set_piece(&t, m.to, BO_PAWN, p->turn);
    return attackers_mask(&t, !p->turn, king, occ_all(&t)) == 0;
}

/* Board._is_safe(king, blockers, move) */
static int is_safe(const bo_pos *p, int king, bb_t blockers, bo_move m) {
    if ((int)m.from == king) {
        if (is_castling_move(p, m)) return 1;
        return attackers_mask(p, !p->turn, m.to, occ_all(p)) == 0;
    }
    if (is_en_passant(p, m)) return ep_is_safe(p, king, m);
    return !(blockers & BB(m.from)) || (RAY[m.from][m.to] & BB(king));
}

/* Board._generate_evasions(king, checkers) */
static void gen_evasions(const bo_pos *p, int king, bb_t checkers, mlist *l) {
    bb_t sliders = checkers & (p->bishops | p->rooks | p->queens);
    bb_t attacked = 0, s = sliders;
    while (s) {
        int c = msb(s);
        s ^= BB(c);
        attacked |= RAY[king][c] & ~BB(c);
    }
    bb_t kt = KING_ATT[king] & ~p->occ[p->turn] & ~attacked;
    while (kt) {
        int to = msb(kt);
        kt ^= BB(to);
        emit(l, king, to, 0);
    }
    int checker = msb(checkers);
    if (BB(checker) == checkers) {
        bb_t target = BETWEEN[king][checker] | checkers;
        gen_pseudo(p, ~p->kings, target, l);
        if (p->ep_square >= 0 && !(BB(p->ep_square) & target)) {
            int last_double = p->ep_square + (p->turn == BO_WHITE ? -8 : 8);
            if (last_double == checker) gen_pseudo_ep(p, ~(bb_t)0, ~(bb_t)0, l);
        }
    }
}

int bo_legal_moves(const bo_pos *p, bo_move *out) {
    bo_rules_init();
    bo_move tmp[BO_MAX_MOVES];
    mlist l = {tmp, 0};
    bb_t kmask = p->kings & p->occ[p->turn];
    if (!kmask) {
        mlist o = {out, 0};
        gen_pseudo(p, ~(bb_t)0, ~(bb_t)0, &o);
        return o.n;
    }
    int king = msb(kmask);
    bb_t blockers = slider_blockers(p, king);
    bb_t checkers = attackers_mask(p, !p->turn, king, occ_all(p));
    if (checkers) gen_evasions(p, king, checkers, &l);
    else gen_pseudo(p, ~(bb_t)0, ~(bb_t)0, &l);
    int n = 0;
    for (int i = 0; i < l.n; i++)
        if (is_safe(p, king, blockers, tmp[i])) out[n++] = tmp[i];
    return n;
}

int bo_has_legal_en_passant(const bo_pos *p) {
    if (p->ep_square < 0) return 0;
    bo_move tmp[4];
    mlist l = {tmp, 0};
    gen_pseudo_ep(p, ~(bb_t)0, ~(bb_t)0, &l);
    if (!l.n) return 0;
    bb_t kmask = p->kings & p->occ[p->turn];
    if (!kmask) return 1;
    int king = msb(kmask);
    for (int i = 0; i < l.n; i++)
        if (ep_is_safe(p, king, tmp[i])) return 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* make move                                                           */
/* ------------------------------------------------------------------ */

int bo_is_zeroing(const bo_pos *p, bo_move m) {
    bb_t touched = BB(m.from) ^ BB(m.to);
    return (touched & p->pawns) != 0 || (touched & p->occ[!p->turn]) != 0;
}

/* Board._reduces_castling_rights(move) */
static int reduces_castling(const bo_pos *p, bo_move m) {
    bb_t cr = p->castling;
    bb_t touched = BB(m.from) ^ BB(m.to);
    return (touched & cr) != 0 ||
           ((cr & BB_RANK_1) && (touched & p->kings & p->occ[BO_WHITE])) ||
           ((cr & BB_RANK_8) && (touched & p->kings & p->occ[BO_BLACK]));
}

int bo_is_irreversible(const bo_pos *p, bo_move m) {
    return bo_is_zeroing(p, m) || reduces_castling(p, m) || bo_has_legal_en_passant(p);
}

void bo_push(bo_pos *p, bo_move m) {
    int ep_old = p->ep_square;
    p->ep_square = -1;
    p->halfmove_clock += 1;
    if (p->turn == BO_BLACK) p->fullmove_number += 1;
    if (bo_is_zeroing(p, m)) p->halfmove_clock = 0;

    int us = p->turn;
    int pt = bo_piece_type_at(p, m.from);
    bb_t from_bb = BB(m.from), to_bb = BB(m.to);
    int castling = 0;
    if (pt == BO_KING) {
        int d = FILE_OF(m.to) - FILE_OF(m.from);
        if (d == 2 || d == -2) castling = 1;
    }
    /* castling rights */
    p->castling &= ~to_bb & ~from_bb;
    if (pt == BO_KING) p->castling &= (us == BO_WHITE) ? ~BB_RANK_1 : ~BB_RANK_8;

    remove_piece(p, m.from);
    if (castling) {
        int base = us == BO_WHITE ? 0 : 56;
        int a_side = FILE_OF(m.to) < FILE_OF(m.from);
        remove_piece(p, base + (a_side ? 0 : 7));
        set_piece(p, base + (a_side ? 2 : 6), BO_KING, us);
        set_piece(p, base + (a_side ? 3 : 5), BO_ROOK, us);
    } else {
        if (pt == BO_PAWN) {
            int diff = (int)m.to - (int)m.from;
            if (diff == 16 && RANK_OF(m.from) == 1) p->ep_square = m.from + 8;
            else if (diff == -16 && RANK_OF(m.from) == 6) p->ep_square = m.from - 8;
            else if ((int)m.to == ep_old && (diff == 7 || diff == 9 || diff == -7 || diff == -9) &&
                     !(occ_all(p) & to_bb)) {
                remove_piece(p, ep_old + (us == BO_WHITE ? -8 : 8));
            }
        }
        if (m.promo) pt = m.promo;
        set_piece(p, m.to, pt, us);
    }
    p->turn = !us;
}

/* ------------------------------------------------------------------ */
/* keys                                                                */
/* ------------------------------------------------------------------ */

void bo_key_of(const bo_pos *p, bo_key *k) {
    k->pawns = p->pawns; k->knights = p->knights; k->bishops = p->bishops;
    k->rooks = p->rooks; k->queens = p->queens; k->kings = p->kings;
    k->occ_w = p->occ[BO_WHITE]; k->occ_b = p->occ[BO_BLACK];
    k->castling = p->castling;
    k->turn = p->turn;
    k->ep = bo_has_legal_en_passant(p) ? p->ep_square : -1;
}

int bo_key_eq(const bo_key *a, const bo_key *b) {
    return a->pawns == b->pawns && a->knights == b->knights && a->bishops == b->bishops &&
           a->rooks == b->rooks && a->queens == b->queens && a->kings == b->kings &&
           a->occ_w == b->occ_w && a->occ_b == b->occ_b && a->castling == b->castling &&
           a->turn == b->turn && a->ep == b->ep;
}

/* ------------------------------------------------------------------ */
/* UCI                                                                 */
/* ------------------------------------------------------------------ */

int bo_move_from_uci(const char *s, bo_move *m) {
    size_t n = strlen(s);
    if (n < 4 || n > 5) return -1;
    if (s[0] < 'a' || s[0] > 'h' || s[2] < 'a' || s[2] > 'h') return -1;
    if (s[1] < '1' || s[1] > '8' || s[3] < '1' || s[3] > '8') return -1;
    m->from = (uint8_t)((s[1] - '1') * 8 + (s[0] - 'a'));
    m->to = (uint8_t)((s[3] - '1') * 8 + (s[2] - 'a'));
    m->promo = 0; m->pad = 0;
    if (n == 5) {
        switch (s[4]) {
        case 'n': m->promo = BO_KNIGHT; break;
        case 'b': m->promo = BO_BISHOP; break;
        case 'r': m->promo = BO_ROOK; break;
        case 'q': m->promo = BO_QUEEN; break;
        default: return -1;
        }
    }
    return 0;
}

void bo_move_to_uci(bo_move m, char *buf) {
    buf[0] = (char)('a' + FILE_OF(m.from)); buf[1] = (char)('1' + RANK_OF(m.from));
    buf[2] = (char)('a' + FILE_OF(m.to)); buf[3] = (char)('1' + RANK_OF(m.to));
    int n = 4;
    if (m.promo) buf[n++] = " pnbrqk"[m.promo];
    buf[n] = 0;
}

/* ------------------------------------------------------------------ */
/* perft                                                               */
/* ------------------------------------------------------------------ */

uint64_t bo_perft(const bo_pos *p, int depth) {
    bo_move mv[BO_MAX_MOVES];
    int n = bo_legal_moves(p, mv);
    if (depth <= 1) return depth == 1 ? (uint64_t)n : 1;
    uint64_t t = 0;
    for (int i = 0; i < n; i++) {
        bo_pos c = *p;
        bo_push(&c, mv[i]);
        t += bo_perft(&c, depth - 1);
    }
    return t;
}

/* ------------------------------------------------------------------ */
/* board with move stack + outcome(claim_draw=True)                    */
/* ------------------------------------------------------------------ */

void bo_stack_reserve(bo_stack *s, int need) {
    if (need <= s->cap) return;
    int cap = s->cap ? s->cap : 64;
    while (cap < need) cap *= 2;
    s->pos = (bo_pos *)realloc(s->pos, sizeof(bo_pos) * (size_t)cap);
    s->key = (bo_key *)realloc(s->key, sizeof(bo_key) * (size_t)cap);
    s->irrev_in = (uint8_t *)realloc(s->irrev_in, (size_t)cap);
    s->cap = cap;
}

void bo_stack_init(bo_stack *s, const bo_pos *start) {
    bo_rules_init();
    memset(s, 0, sizeof(*s));
    bo_stack_reserve(s, 64);
    s->pos[0] = *start;
    bo_key_of(start, &s->key[0]);
    s->irrev_in[0] = 1;
    s->n = 1;
}

void bo_stack_free(bo_stack *s) {
    free(s->pos); free(s->key); free(s->irrev_in);
    memset(s, 0, sizeof(*s));
}

void bo_stack_push(bo_stack *s, bo_move m) {
    bo_stack_reserve(s, s->n + 1);
    const bo_pos *cur = &s->pos[s->n - 1];
    s->irrev_in[s->n] = (uint8_t)bo_is_irreversible(cur, m);
    s->pos[s->n] = *cur;
    bo_push(&s->pos[s->n], m);
    bo_key_of(&s->pos[s->n], &s->key[s->n]);
    s->n++;
}

void bo_stack_pop(bo_stack *s) { if (s->n > 1) s->n--; }

void bo_stack_copy(bo_stack *dst, const bo_stack *src) {
    memset(dst, 0, sizeof(*dst));
    bo_stack_reserve(dst, src->n + 16);
    memcpy(dst->pos, src->pos, sizeof(bo_pos) * (size_t)src->n);
    memcpy(dst->key, src->key, sizeof(bo_key) * (size_t)src->n);
    memcpy(dst->irrev_in, src->irrev_in, (size_t)src->n);
    dst->n = src->n;
}

/* occurrences of `k` among the positions python-chess would visit when it
 * pops moves back to (and excluding the position before) the last
 * irreversible move; the current position itself is NOT counted here. */
static int chain_count(const bo_stack *s, const bo_key *k) {
    int c = 0;
    for (int i = s->n - 1; i > 0; i--) {
        if (s->irrev_in[i]) break;
        if (bo_key_eq(&s->key[i - 1], k)) c++;
    }
    return c;
}

int bo_termination_claim_draw(bo_stack *s) {
    const bo_pos *p = bo_stack_top(s);
    bo_move mv[BO_MAX_MOVES];
    int n = bo_legal_moves(p, mv);
    if (n == 0 && bo_is_check(p)) return 1;
    if (bo_is_insufficient_material(p)) return 2;
    if (n == 0) return 3;
    if (p->halfmove_clock >= 150) return 4;
    const bo_key *cur = &s->key[s->n - 1];
    int occurrences = 1 + chain_count(s, cur);
    if (occurrences >= 5) return 5;
    /* can_claim_fifty_moves */
    if (p->halfmove_clock >= 100) return 6;
    if (p->halfmove_clock >= 99) {
        for (int i = 0; i < n; i++) {
            if (bo_is_zeroing(p, mv[i])) continue;
            bo_pos c = *p;
            bo_push(&c, mv[i]);
            bo_move t[BO_MAX_MOVES];
            if (c.halfmove_clock >= 100 && bo_legal_moves(&c, t) > 0) return 6;
        }
    }
    /* can_claim_threefold_repetition */
    if (occurrences >= 3) return 7;
    for (int i = 0; i < n; i++) {
        bo_pos c = *p;
        bo_push(&c, mv[i]);
        bo_key ck;
        bo_key_of(&c, &ck);
        /* Counter holds the chain incl. the current position */
        int cnt = chain_count(s, &ck) + (bo_key_eq(cur, &ck) ? 1 : 0);
        if (cnt >= 2) return 7;
    }
    return 0;
}

int bo_outcome_claim_draw(bo_stack *s) {
    int t = bo_termination_claim_draw(s);
    if (t == 0) return BO_ONGOING;
    return t == 1 ? BO_CHECKMATE : BO_DRAW;
}

int bo_is_insufficient_material(const bo_pos *p) {
    for (int color = 0; color < 2; color++) {
        bb_t own = p->occ[color];
        if (own & (p->pawns | p->rooks | p->queens)) return 0;
        if (own & p->knights) {
            if (!(popcnt(own) <= 2 && !(p->occ[!color] & ~p->kings & ~p->queens))) return 0;
            continue;
        }
        if (own & p->bishops) {
            int same = !(p->bishops & BB_DARK) || !(p->bishops & BB_LIGHT);
            if (!(same && !p->pawns && !p->knights)) return 0;
            continue;
        }
    }
    return 1;
}
