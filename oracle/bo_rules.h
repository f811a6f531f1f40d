/*
 * oracle/bo_rules.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the chess rules the reference takes from the
 * third-party package python-chess (pinned chess==1.11.2,
 * /root/reference/train_requirements.txt:2), which is NOT vendored under
 * /root/reference and NOT installed in this image.  The published algorithm of
 * python-chess is restated here from its documented behaviour:
 *   - legal move generation IN python-chess's generation order
 *     (Board.generate_legal_moves / generate_pseudo_legal_moves /
 *      _generate_evasions / generate_castling_moves);
 *   - Board.push (ep square after every double push, half-move clock,
 *     full-move number, castling-right updates);
 *   - Board._transposition_key (ep square only when an ep capture is legal);
 *   - Board.outcome(claim_draw=True) ordering: checkmate, insufficient
 *     material, stalemate, 75-move, fivefold, claimable 50-move, claimable
 *     threefold (incl. the one-ply lookahead);
 *   - Board.is_irreversible (zeroing, castling-right reducing, legal ep).
 * Call sites of these in the reference: mcts.py:36-37,66-67,152,186,191,260,292;
 * utils.py:78,156,191-215,387-389; self_play.py:91-184.
 *
 * PARITY STATUS: the *set* of legal moves is pinned by public perft
 * known-answers (tests/test_oracle_rules.py).  Move ORDER and the claim_draw
 * edge cases are restated from python-chess's published source from memory and
 * are "parity unpinned" against python-chess itself.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#ifndef BO_RULES_H
#define BO_RULES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t bb_t;

/* python-chess piece types */
enum { BO_PAWN = 1, BO_KNIGHT = 2, BO_BISHOP = 3, BO_ROOK = 4, BO_QUEEN = 5, BO_KING = 6 };
/* python-chess colors: WHITE = True = 1, BLACK = False = 0 */
enum { BO_BLACK = 0, BO_WHITE = 1 };

typedef struct {
    bb_t pawns, knights, bishops, rooks, queens, kings;
    bb_t occ[2];        /* occ[BO_WHITE], occ[BO_BLACK] */
    bb_t castling;      /* rook squares that still carry a castling right (clean) */
    int32_t turn;       /* BO_WHITE / BO_BLACK */
    int32_t ep_square;  /* raw ep square (set after every double push) or -1 */
    int32_t halfmove_clock;
    int32_t fullmove_number;
} bo_pos;

typedef struct {
    uint8_t from, to, promo, pad; /* promo: 0 or BO_KNIGHT..BO_QUEEN */
} bo_move;

/* exact transposition key, python-chess Board._transposition_key() */
typedef struct {
    bb_t pawns, knights, bishops, rooks, queens, kings, occ_w, occ_b;
    bb_t castling;
    int32_t turn;
    int32_t ep; /* ep square if an ep capture is legal, else -1 */
} bo_key;

#define BO_MAX_MOVES 256

void bo_rules_init(void);

/* FEN -> position.  returns 0 on success. */
int bo_pos_from_fen(const char *fen, bo_pos *out);
void bo_pos_startpos(bo_pos *out);
/* position -> FEN (python-chess Board.fen(): ep field only when legal). */
void bo_pos_to_fen(const bo_pos *p, char *buf, int buflen);

/* legal moves in python-chess generation order; returns count */
int bo_legal_moves(const bo_pos *p, bo_move *out);
int bo_is_check(const bo_pos *p);
int bo_has_legal_en_passant(const bo_pos *p);
int bo_is_insufficient_material(const bo_pos *p);

/* Board.push for a (pseudo-)legal move */
void bo_push(bo_pos *p, bo_move m);
/* Board.is_zeroing / Board.is_irreversible evaluated on the position BEFORE m */
int bo_is_zeroing(const bo_pos *p, bo_move m);
int bo_is_irreversible(const bo_pos *p, bo_move m);

void bo_key_of(const bo_pos *p, bo_key *k);
int bo_key_eq(const bo_key *a, const bo_key *b);

/* UCI text <-> move ("e2e4", "e7e8q"); parse returns 0 on success */
int bo_move_from_uci(const char *s, bo_move *m);
void bo_move_to_uci(bo_move m, char *buf);

/* piece type at square (0 if empty) and its color */
int bo_piece_type_at(const bo_pos *p, int sq);
int bo_color_at(const bo_pos *p, int sq);

uint64_t bo_perft(const bo_pos *p, int depth);

/*
 * A board with its move stack (python-chess Board incl. move_stack), as far as
 * the draw rules need it: the chain of positions reached so far, and for each
 * position whether the move that led to it was irreversible.
 */
typedef struct {
    bo_pos *pos;       /* pos[0..n-1], pos[n-1] is the current position */
    bo_key *key;       /* key[i] = transposition key of pos[i] */
    uint8_t *irrev_in; /* irrev_in[i]: move pos[i-1]->pos[i] was irreversible (irrev_in[0]=1) */
    int n, cap;
} bo_stack;

void bo_stack_init(bo_stack *s, const bo_pos *start);
void bo_stack_free(bo_stack *s);
void bo_stack_reserve(bo_stack *s, int need);
void bo_stack_push(bo_stack *s, bo_move m);
void bo_stack_pop(bo_stack *s);
void bo_stack_copy(bo_stack *dst, const bo_stack *src);
static inline const bo_pos *bo_stack_top(const bo_stack *s) { return &s->pos[s->n - 1]; }

/* outcome(claim_draw=True): returns 0 = game not over, 1 = checkmate (side to
 * move is mated), 2 = draw (any of the draw terminations). */
enum { BO_ONGOING = 0, BO_CHECKMATE = 1, BO_DRAW = 2 };
int bo_outcome_claim_draw(bo_stack *s);
/* detailed termination for tests: 0 none,1 mate,2 insufficient,3 stalemate,
 * 4 seventyfive,5 fivefold,6 fifty-claim,7 threefold-claim */
int bo_termination_claim_draw(bo_stack *s);

#ifdef __cplusplus
}
#endif
#endif
