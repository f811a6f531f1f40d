// betaone_amd/csrc/bo_chess.h -- chess rules for the gfx950 tree kernels (device code).
//
// What this replaces in the reference: every python-chess call on the self-play path
// (SURVEY.md section 8c): Board.legal_moves in python-chess generation order (mcts.py:186,191,260,292),
// Board.push (mcts.py:67, self_play.py:171), Board._transposition_key (mcts.py:37, utils.py:78-103),
// Board.is_game_over(claim_draw=True)/result (mcts.py:152, utils.py:385-396), plus the reference's own
// move codec utils.move_to_index / index_to_move (utils.py:221-365).
//
// MI355X-first design, not a translation of python-chess:
//   * a position is 8 bitboards + one flag word (80 B, one cache line and a quarter);
//   * move generation runs ONE LANE PER SQUARE of a 64-wide wavefront: every lane computes the
//     rook/bishop rays of its own square once (hyperbola quintessence, v_bfrev for the reversed
//     half), and from those rays alone derives (a) whether its square is attacked (ballot -> the
//     opponent's attack map), (b) whether its piece is pinned, (c) its piece's legal targets;
//   * python-chess's generation order (pieces by from-square high->low, castling, pawn captures,
//     pushes, double pushes, en passant; king moves first when in check) is reproduced with two
//     packed wave prefix sums instead of a serial loop, so the ordered move list is written in
//     parallel;
//   * the draw rules that need the move stack (fivefold, claimable threefold incl. one-ply
//     lookahead, claimable fifty-move) run one lane per legal move.
#pragma once
#include "bo_wave.h"

#define BO_MAX_MOVES 256
#define BO_NUM_ACTIONS 4672

// ---- position ---------------------------------------------------------------------------------
enum { BB_P = 0, BB_N = 1, BB_B = 2, BB_R = 3, BB_Q = 4, BB_K = 5, BB_WHITE = 6, BB_BLACK = 7 };

struct DPos {
    uint64_t bb[8];
    uint32_t flags;  // see F_* below
    int32_t halfmove;
    int32_t fullmove;
    uint32_t khash;  // 32-bit mix of the exact transposition key (filter only; equality is exact)
};
// flags layout
#define F_TURN 0x1u          // 1 = white to move
#define F_CASTLE_SHIFT 1     // bit1 K(h1) bit2 Q(a1) bit3 k(h8) bit4 q(a8)
#define F_CASTLE_MASK 0x1Eu
#define F_EP_SHIFT 8         // bits 8..14 : raw ep square + 1 (0 = none)  [Board.ep_square]
#define F_EP_MASK (0x7Fu << 8)
#define F_IRREV 0x8000u      // the move that led to this position was irreversible
#define F_EPKEY_SHIFT 16     // bits 16..22: ep square + 1 if an ep capture is LEGAL (key component)
#define F_EPKEY_MASK (0x7Fu << 16)
#define F_KEY_MASK (F_TURN | F_CASTLE_MASK | F_EPKEY_MASK)

#define BIT(sq) (1ULL << (sq))
#define FILE_A 0x0101010101010101ULL
#define FILE_H 0x8080808080808080ULL
#define FILE_AB 0x0303030303030303ULL
#define FILE_GH 0xC0C0C0C0C0C0C0C0ULL
#define RANK_1 0x00000000000000FFULL
#define RANK_8 0xFF00000000000000ULL
#define DARK_SQ 0xaa55aa55aa55aa55ULL
#define LIGHT_SQ 0x55aa55aa55aa55aaULL

typedef uint16_t bo_mv;  // from | to<<6 | promo<<12   (promo: 0 or python-chess piece type 2..5)
#define MV(from, to, promo) ((bo_mv)((from) | ((to) << 6) | ((promo) << 12)))
#define MV_FROM(m) ((int)((m) & 63))
#define MV_TO(m) ((int)(((m) >> 6) & 63))
#define MV_PROMO(m) ((int)(((m) >> 12) & 7))

BO_DEV int pos_turn(const DPos &p) { return (int)(p.flags & F_TURN); }
BO_DEV int pos_ep(const DPos &p) { return (int)((p.flags & F_EP_MASK) >> F_EP_SHIFT) - 1; }
BO_DEV uint64_t pos_our(const DPos &p) { return p.bb[pos_turn(p) ? BB_WHITE : BB_BLACK]; }
BO_DEV uint64_t pos_their(const DPos &p) { return p.bb[pos_turn(p) ? BB_BLACK : BB_WHITE]; }
BO_DEV uint64_t pos_all(const DPos &p) { return p.bb[BB_WHITE] | p.bb[BB_BLACK]; }

// ---- line geometry (pure ALU, no tables) ------------------------------------------------------------
BO_DEV uint64_t file_mask(int s) { return FILE_A << (s & 7); }
BO_DEV uint64_t rank_mask(int s) { return RANK_1 << (s & 56); }
BO_DEV uint64_t diag_mask(int s) {  // a1-h8 direction
    int d = (s >> 3) - (s & 7);
    return d >= 0 ? (0x8040201008040201ULL << (8 * d)) : (0x8040201008040201ULL >> (8 * -d));
}
BO_DEV uint64_t anti_mask(int s) {  // h1-a8 direction
    int d = (s >> 3) + (s & 7) - 7;
    return d >= 0 ? (0x0102040810204080ULL << (8 * d)) : (0x0102040810204080ULL >> (8 * -d));
}
// sliding attacks along one line through s (hyperbola quintessence)
BO_DEV uint64_t line_att(uint64_t occ, int s, uint64_t m) {
    uint64_t b = BIT(s), o = occ & m;
    uint64_t fwd = o - 2 * b;
    uint64_t rev = bo_bitrev64(bo_bitrev64(o) - 2 * bo_bitrev64(b));
    return (fwd ^ rev) & m;
}
BO_DEV uint64_t rook_att(int s, uint64_t occ) { return line_att(occ, s, file_mask(s)) | line_att(occ, s, rank_mask(s)); }
BO_DEV uint64_t bishop_att(int s, uint64_t occ) { return line_att(occ, s, diag_mask(s)) | line_att(occ, s, anti_mask(s)); }
BO_DEV uint64_t knight_att(int s) {
    uint64_t b = BIT(s);
    return ((b << 17) & ~FILE_A) | ((b << 15) & ~FILE_H) | ((b << 10) & ~FILE_AB) | ((b << 6) & ~FILE_GH) |
           ((b >> 17) & ~FILE_H) | ((b >> 15) & ~FILE_A) | ((b >> 10) & ~FILE_GH) | ((b >> 6) & ~FILE_AB);
}
BO_DEV uint64_t king_att(int s) {
    uint64_t b = BIT(s);
    uint64_t h = ((b << 1) & ~FILE_A) | ((b >> 1) & ~FILE_H);
    uint64_t r = b | h;
    return h | (r << 8) | (r >> 8);
}
// squares attacked by a pawn of `color` (1 white) standing on s
BO_DEV uint64_t pawn_att(int color, int s) {
    uint64_t b = BIT(s);
    return color ? (((b << 7) & ~FILE_H) | ((b << 9) & ~FILE_A)) : (((b >> 7) & ~FILE_A) | ((b >> 9) & ~FILE_H));
}
// whole line through a and b (edge to edge) or 0 when not aligned   [python-chess ray()]
BO_DEV uint64_t line_through(int a, int b) {
    if (a == b) return 0;
    if ((a & 7) == (b & 7)) return file_mask(a);
    if ((a >> 3) == (b >> 3)) return rank_mask(a);
    if ((a >> 3) - (a & 7) == (b >> 3) - (b & 7)) return diag_mask(a);
    if ((a >> 3) + (a & 7) == (b >> 3) + (b & 7)) return anti_mask(a);
    return 0;
}
// squares strictly between a and b when aligned, else 0        [python-chess between()]
BO_DEV uint64_t between_bb(int a, int b) {
    uint64_t l = line_through(a, b);
    int lo = a < b ? a : b, hi = a < b ? b : a;
    uint64_t span = (BIT(hi) - 1) & ~(BIT(lo) | (BIT(lo) - 1));
    return l & span;
}

// ---- per-position helpers (scalar style: one lane, one position) ---------------------------------
BO_DEV int piece_type_at(const DPos &p, int s) {  // 0 none, 1..6 python-chess piece types
    uint64_t b = BIT(s);
    int t = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) t = (p.bb[i] & b) ? i + 1 : t;
    return t;
}
// opponents of `us` attacking sq under occupancy `occ`, ignoring pieces in `gone`
BO_DEV uint64_t attackers_of(const DPos &p, int us, int sq, uint64_t occ, uint64_t gone) {
    uint64_t their = p.bb[us ? BB_BLACK : BB_WHITE] & ~gone;
    uint64_t a = (rook_att(sq, occ) & (p.bb[BB_R] | p.bb[BB_Q])) | (bishop_att(sq, occ) & (p.bb[BB_B] | p.bb[BB_Q])) |
                 (knight_att(sq) & p.bb[BB_N]) | (king_att(sq) & p.bb[BB_K]) | (pawn_att(us, sq) & p.bb[BB_P]);
    return a & their;
}
// legality of the ep capture from `from` (the capturing pawn is known to attack the ep square)
BO_DEV bool ep_capture_safe(const DPos &p, int from) {
    int us = pos_turn(p), ep = pos_ep(p);
    int capsq = ep + (us ? -8 : 8);
    uint64_t kbb = p.bb[BB_K] & pos_our(p);
    if (!kbb) return true;
    uint64_t occ = (pos_all(p) ^ BIT(from) ^ BIT(capsq)) | BIT(ep);
    return attackers_of(p, us, bo_lsb64(kbb), occ, BIT(capsq)) == 0;
}
// Board.has_legal_en_passant()
BO_DEV bool has_legal_ep(const DPos &p) {
    int ep = pos_ep(p);
    if (ep < 0) return false;
    int us = pos_turn(p);
    if (pos_all(p) & BIT(ep)) return false;
    uint64_t rank = us ? (RANK_1 << 32) : (RANK_1 << 24);
    uint64_t cap = p.bb[BB_P] & pos_our(p) & pawn_att(!us, ep) & rank;
    while (cap) {
        int c = bo_lsb64(cap);
        cap &= cap - 1;
        if (ep_capture_safe(p, c)) return true;
    }
    return false;
}
BO_DEV uint32_t key_hash(const DPos &p) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ (uint64_t)(p.flags & F_KEY_MASK);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        h ^= p.bb[i];
        h *= 0xBF58476D1CE4E5B9ULL;
        h ^= h >> 29;
    }
    return (uint32_t)(h ^ (h >> 32));
}
// exact transposition-key equality (python-chess Board._transposition_key())
BO_DEV bool key_equal(const DPos &a, const DPos &b) {
    bool e = ((a.flags ^ b.flags) & F_KEY_MASK) == 0;
#pragma unroll
    for (int i = 0; i < 8; i++) e = e && (a.bb[i] == b.bb[i]);
    return e;
}
// Board.is_zeroing(move)
BO_DEV bool is_zeroing(const DPos &p, bo_mv m) {
    uint64_t touched = BIT(MV_FROM(m)) ^ BIT(MV_TO(m));
    return (touched & p.bb[BB_P]) != 0 || (touched & pos_their(p)) != 0;
}
BO_DEV uint32_t castle_bits_touched(uint64_t sqs) {  // castling-right bits whose rook square is in sqs
    return (uint32_t)((((sqs >> 7) & 1) << 1) | (((sqs >> 0) & 1) << 2) | (((sqs >> 63) & 1) << 3) | (((sqs >> 56) & 1) << 4));
}
// Board.is_irreversible(move) evaluated on the position BEFORE the move
BO_DEV bool is_irreversible(const DPos &p, bo_mv m) {
    uint64_t touched = BIT(MV_FROM(m)) ^ BIT(MV_TO(m));
    uint32_t cr = p.flags & F_CASTLE_MASK;
    bool reduces = (castle_bits_touched(touched) & cr) != 0 ||
                   ((cr & 0x06u) && (touched & p.bb[BB_K] & p.bb[BB_WHITE])) ||
                   ((cr & 0x18u) && (touched & p.bb[BB_K] & p.bb[BB_BLACK]));
    return is_zeroing(p, m) || reduces || (p.flags & F_EPKEY_MASK) != 0;
}
// Board.push(move): the child position, complete with its key fields
BO_DEV DPos make_move(const DPos &p, bo_mv m) {
    DPos c = p;
    int from = MV_FROM(m), to = MV_TO(m), promo = MV_PROMO(m);
    int us = pos_turn(p);
    uint64_t fb = BIT(from), tb = BIT(to);
    int pt = piece_type_at(p, from);
    bool zero = is_zeroing(p, m);
    c.halfmove = zero ? 0 : p.halfmove + 1;
    c.fullmove = p.fullmove + (us ? 0 : 1);
    uint32_t cr = p.flags & F_CASTLE_MASK;
    cr &= ~castle_bits_touched(fb | tb);
    if (pt == 6) cr &= us ? ~0x06u : ~0x18u;
    int ep_old = pos_ep(p), ep_new = -1;
    int ourc = us ? BB_WHITE : BB_BLACK, theirc = us ? BB_BLACK : BB_WHITE;
    // lift the moving piece
#pragma unroll
    for (int i = 0; i < 6; i++) c.bb[i] &= ~fb;
    c.bb[ourc] &= ~fb;
    int df = (to & 7) - (from & 7);
    if (pt == 6 && (df == 2 || df == -2)) {  // castling (e1g1 / e1c1 form)
        int base = us ? 0 : 56;
        int rf = df < 0 ? base : base + 7, rt = df < 0 ? base + 3 : base + 5;
        c.bb[BB_R] = (c.bb[BB_R] & ~BIT(rf)) | BIT(rt);
        c.bb[ourc] = (c.bb[ourc] & ~BIT(rf)) | BIT(rt) | tb;
        c.bb[BB_K] |= tb;
    } else {
        if (pt == 1) {
            int diff = to - from;
            if (diff == 16 && (from >> 3) == 1) ep_new = from + 8;
            else if (diff == -16 && (from >> 3) == 6) ep_new = from - 8;
            else if (to == ep_old && (diff == 7 || diff == 9 || diff == -7 || diff == -9) && !(pos_all(p) & tb)) {
                uint64_t cb = BIT(ep_old + (us ? -8 : 8));
                c.bb[BB_P] &= ~cb;
                c.bb[theirc] &= ~cb;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) c.bb[i] &= ~tb;  // capture
        c.bb[theirc] &= ~tb;
        int np = promo ? promo : pt;
        c.bb[np - 1] |= tb;
        c.bb[ourc] |= tb;
    }
    c.flags = (us ? 0u : F_TURN) | cr | ((uint32_t)(ep_new + 1) << F_EP_SHIFT);
    if (is_irreversible(p, m)) c.flags |= F_IRREV;
    if (ep_new >= 0 && has_legal_ep(c)) c.flags |= (uint32_t)(ep_new + 1) << F_EPKEY_SHIFT;
    c.khash = key_hash(c);
    return c;
}
// Board.is_insufficient_material()
BO_DEV bool insufficient_material(const DPos &p) {
    bool all_ok = true;
#pragma unroll
    for (int color = 0; color < 2; color++) {
        uint64_t own = p.bb[color ? BB_WHITE : BB_BLACK], opp = p.bb[color ? BB_BLACK : BB_WHITE];
        bool ok;
        if (own & (p.bb[BB_P] | p.bb[BB_R] | p.bb[BB_Q])) ok = false;
        else if (own & p.bb[BB_N]) ok = bo_popc64(own) <= 2 && !(opp & ~p.bb[BB_K] & ~p.bb[BB_Q]);
        else if (own & p.bb[BB_B]) {
            bool same = !(p.bb[BB_B] & DARK_SQ) || !(p.bb[BB_B] & LIGHT_SQ);
            ok = same && !p.bb[BB_P] && !p.bb[BB_N];
        } else ok = true;
        all_ok = all_ok && ok;
    }
    return all_ok;
}

// ---- move <-> action index (utils.py:221-365) ------------------------------------------------------
BO_DEV int move_to_index(bo_mv m) {
    int from = MV_FROM(m), to = MV_TO(m), promo = MV_PROMO(m);
    int fr = from >> 3, ff = from & 7, dr = (to >> 3) - fr, df = (to & 7) - ff;
    if (promo && promo != 5)  // under-promotion planes 64..72: piece (N,B,R) x file direction (left, straight, right)
        return from * 73 + 64 + (promo - 2) * 3 + (df + 1);
    int adr = dr < 0 ? -dr : dr, adf = df < 0 ? -df : df;
    if ((adr == 1 && adf == 2) || (adr == 2 && adf == 1)) {
        // KNIGHT_DIRECTIONS (2,1),(1,2),(-1,2),(-2,1),(-2,-1),(-1,-2),(1,-2),(2,-1)
        int k = dr == 2 ? (df == 1 ? 0 : 7) : dr == 1 ? (df == 2 ? 1 : 6) : dr == -1 ? (df == 2 ? 2 : 5) : (df == 1 ? 3 : 4);
        return from * 73 + 56 + k;
    }
    int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
    // QUEEN_DIRECTIONS N,NE,E,SE,S,SW,W,NW as (d_rank, d_file)
    int dir = sr == 1 ? (sf == 0 ? 0 : sf == 1 ? 1 : 7) : sr == 0 ? (sf == 1 ? 2 : 6) : (sf == 1 ? 3 : sf == 0 ? 4 : 5);
    int dist = adr > adf ? adr : adf;
    return from * 73 + dir * 7 + (dist - 1);
}
// returns false where the reference raises ValueError
BO_DEV bool index_to_move(int index, const DPos &p, bo_mv *out) {
    if (index < 0 || index >= BO_NUM_ACTIONS) return false;
    int from = index / 73, plane = index % 73, fr = from >> 3, ff = from & 7;
    const int qdr[8] = {1, 1, 0, -1, -1, -1, 0, 1}, qdf[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    const int ndr[8] = {2, 1, -1, -2, -2, -1, 1, 2}, ndf[8] = {1, 2, 2, 1, -1, -2, -2, -1};
    int tr, tf, promo = 0;
    if (plane < 56) {
        int dir = plane / 7, dist = plane % 7 + 1;
        tr = fr + qdr[dir] * dist;
        tf = ff + qdf[dir] * dist;
        if (tr < 0 || tr > 7 || tf < 0 || tf > 7) return false;
        if (p.bb[BB_P] & BIT(from)) {
            bool white = (p.bb[BB_WHITE] & BIT(from)) != 0;
            if ((white && fr == 6 && tr == 7) || (!white && fr == 1 && tr == 0)) promo = 5;
        }
    } else if (plane < 64) {
        tr = fr + ndr[plane - 56];
        tf = ff + ndf[plane - 56];
        if (tr < 0 || tr > 7 || tf < 0 || tf > 7) return false;
    } else {
        int off = plane - 64, piece = off / 3, dir = off % 3;
        if (!(p.bb[BB_P] & BIT(from))) return false;
        bool white = (p.bb[BB_WHITE] & BIT(from)) != 0;
        int dr;
        if (white && fr == 6) dr = 1;
        else if (!white && fr == 1) dr = -1;
        else return false;
        tr = fr + dr;
        tf = ff + (dir - 1);
        if (tr < 0 || tr > 7 || tf < 0 || tf > 7) return false;
        promo = 2 + piece;
    }
    *out = MV(from, tr * 8 + tf, promo);
    return true;
}

// ---- wave-parallel legal move generation ------------------------------------------------------------
// All 64 lanes call this with the SAME position.  Writes the legal moves, in python-chess
// generation order, to out[0..n) (LDS or global) and returns n; *in_check_out = side to move in check.
BO_DEV_NOINLINE int bo_movegen(const DPos &P, bo_mv *out, bool *in_check_out) {
    const int s = bo_lane();
    const uint64_t bit = BIT(s);
    const int us = pos_turn(P);
    const uint64_t our = pos_our(P), their = pos_their(P), all = our | their;
    const uint64_t kbb = P.bb[BB_K] & our;
    const int ksq = kbb ? bo_lsb64(kbb) : 0;
    const uint64_t oppRQ = (P.bb[BB_R] | P.bb[BB_Q]) & their, oppBQ = (P.bb[BB_B] | P.bb[BB_Q]) & their;

    // rays of this lane's square, once
    const uint64_t R = rook_att(s, all), B = bishop_att(s, all);
    const uint64_t KN = knight_att(s), KG = king_att(s);
    // is my square attacked by the opponent?  (ballot -> opponent attack map, lane index == square)
    const uint64_t att_s = (R & oppRQ) | (B & oppBQ) | (KN & P.bb[BB_N] & their) | (KG & P.bb[BB_K] & their) |
                           (pawn_att(us, s) & P.bb[BB_P] & their);
    const uint64_t attacked = bo_ballot(att_s != 0);
    const uint64_t checkers = kbb ? bo_shfl_u64(att_s, ksq) : 0;
    const int nchk = bo_popc64(checkers);

    // evasion masks (wave-uniform)
    uint64_t to_mask = ~0ULL, xray = 0;
    if (nchk) {
        uint64_t c = checkers;
        while (c) {
            int cs = bo_lsb64(c);
            c &= c - 1;
            if (BIT(cs) & (P.bb[BB_B] | P.bb[BB_R] | P.bb[BB_Q])) xray |= line_through(ksq, cs) & ~BIT(cs);
        }
        to_mask = nchk == 1 ? (between_bb(ksq, bo_lsb64(checkers)) | checkers) : 0;
    }

    // pin: my piece, our king and an enemy slider on one line with nothing else between
    const bool own = (our & bit) != 0;
    uint64_t pin_line = ~0ULL;
    if (own && kbb && s != ksq) {
        uint64_t L = line_through(s, ksq);
        if (L) {
            bool rooklike = (s & 7) == (ksq & 7) || (s >> 3) == (ksq >> 3);
            uint64_t rays = rooklike ? R : B;
            if (rays & kbb) {
                uint64_t far = rays & L & ~between_bb(s, ksq) & ~kbb;
                if (far & all & (rooklike ? oppRQ : oppBQ)) pin_line = L;
            }
        }
    }

    // targets by category
    uint64_t tK = 0, tA = 0, tC = 0, t1 = 0, t2 = 0;
    bool epc = false;
    int nB = 0;  // castling moves (lane ksq)
    bool castle_k = false, castle_q = false;
    if (own) {
        if (P.bb[BB_P] & bit) {
            tC = pawn_att(us, s) & their & to_mask & pin_line;
            uint64_t one = us ? bit << 8 : bit >> 8;
            one &= ~all;
            uint64_t two = 0;
            if (one && (s >> 3) == (us ? 1 : 6)) two = (us ? one << 8 : one >> 8) & ~all;
            t1 = one & to_mask & pin_line;
            t2 = two & to_mask & pin_line;
            int ep = pos_ep(P);
            if (ep >= 0 && (pawn_att(us, s) & BIT(ep)) && (s >> 3) == (us ? 4 : 3) && !(all & BIT(ep))) {
                bool gen = nchk == 0;
                if (nchk == 1) gen = (BIT(ep) & to_mask) != 0 || bo_lsb64(checkers) == ep + (us ? -8 : 8);
                if (gen) epc = ep_capture_safe(P, s);
            }
        } else if (s == ksq && kbb) {
            uint64_t t = KG & ~our & ~attacked & ~xray;
            if (nchk) tK = t;
            else {
                tA = t;
                uint32_t cr = P.flags & F_CASTLE_MASK;
                int base = us ? 0 : 56;
                bool rk = (cr & (us ? 0x02u : 0x08u)) != 0, rq = (cr & (us ? 0x04u : 0x10u)) != 0;
                uint64_t fg = (BIT(5) | BIT(6)) << base, bcd = (BIT(1) | BIT(2) | BIT(3)) << base, cd = (BIT(2) | BIT(3)) << base;
                castle_k = rk && !(all & fg) && !(attacked & fg);
                castle_q = rq && !(all & bcd) && !(attacked & cd);
                nB = (int)castle_k + (int)castle_q;
            }
        } else {
            uint64_t t = 0;
            if (P.bb[BB_N] & bit) t = KN;
            if ((P.bb[BB_B] | P.bb[BB_Q]) & bit) t |= B;
            if ((P.bb[BB_R] | P.bb[BB_Q]) & bit) t |= R;
            tA = t & ~our & to_mask & pin_line;
        }
    }
    const uint64_t promo_rank = us ? RANK_8 : RANK_1;
    const int cK = bo_popc64(tK), cA = bo_popc64(tA);
    const int cC = bo_popc64(tC & ~promo_rank) + 4 * bo_popc64(tC & promo_rank);
    const int c1 = bo_popc64(t1 & ~promo_rank) + 4 * bo_popc64(t1 & promo_rank);
    const int c2 = bo_popc64(t2), cF = epc ? 1 : 0;

    // python-chess order = category-major, from-square descending: two packed descending scans
    const int v1 = cK | (cA << 8) | (cC << 16) | (c1 << 24);
    const int v2 = c2 | (cF << 8) | (nB << 16);
    const int i1 = bo_wave_scan_desc(v1), i2 = bo_wave_scan_desc(v2);
    const int tot1 = bo_shfl(i1, 0), tot2 = bo_shfl(i2, 0);
    const int nK = tot1 & 255, nA = (tot1 >> 8) & 255, nC = (tot1 >> 16) & 255, n1 = (tot1 >> 24) & 255;
    const int n2 = tot2 & 255, nF = (tot2 >> 8) & 255, nBt = (tot2 >> 16) & 255;
    const int bA = nK, bB = bA + nA, bC = bB + nBt, b1 = bC + nC, b2 = b1 + n1, bF = b2 + n2;
    const int total = bF + nF;

    int o = (i1 & 255) - cK;  // king moves (evasions first)
    for (uint64_t t = tK; t;) { int to = bo_msb64(t); t ^= BIT(to); out[o++] = MV(s, to, 0); }
    o = bA + ((i1 >> 8) & 255) - cA;
    for (uint64_t t = tA; t;) { int to = bo_msb64(t); t ^= BIT(to); out[o++] = MV(s, to, 0); }
    if (nB) {
        o = bB;
        if (castle_k) out[o++] = MV(s, s + 2, 0);
        if (castle_q) out[o++] = MV(s, s - 2, 0);
    }
    o = bC + ((i1 >> 16) & 255) - cC;
    for (uint64_t t = tC; t;) {
        int to = bo_msb64(t);
        t ^= BIT(to);
        if (BIT(to) & promo_rank) { out[o++] = MV(s, to, 5); out[o++] = MV(s, to, 4); out[o++] = MV(s, to, 3); out[o++] = MV(s, to, 2); }
        else out[o++] = MV(s, to, 0);
    }
    o = b1 + ((i1 >> 24) & 255) - c1;
    if (t1) {
        int to = bo_lsb64(t1);
        if (t1 & promo_rank) { out[o++] = MV(s, to, 5); out[o++] = MV(s, to, 4); out[o++] = MV(s, to, 3); out[o++] = MV(s, to, 2); }
        else out[o++] = MV(s, to, 0);
    }
    if (t2) out[b2 + (i2 & 255) - c2] = MV(s, bo_lsb64(t2), 0);
    if (epc) out[bF + ((i2 >> 8) & 255) - cF] = MV(s, pos_ep(P), 0);
    bo_sync();
    *in_check_out = nchk > 0;
    return total;
}
