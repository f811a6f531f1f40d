/*
 * oracle/bo_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * C restatement of the reference's self-play rollout path:
 *   mcts.py:19-152   MCTSNode (expand / select_child / update / update_recursive)
 *   mcts.py:155-295  run_mcts, _evaluate_batch
 *   self_play.py:84-216  run_self_play_game (game loop, z sign, final encodings)
 *   utils.py         codec (bo_codec.c)
 * in the dtype regime R3 of SURVEY.md section 8: net outputs are float32,
 * softmax is float32, and every tree operation is ONE IEEE-754 binary32
 * operation (NumPy-2 scalar promotion), compiled with -ffp-contract=off.
 *
 * The network and NumPy's global RNG are NOT restated: they are reached through
 * callbacks so that tests can plug torch-CPU / numpy.random.RandomState (the
 * same objects the reference itself calls).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (betaone_amd/) never links or imports it.
 */
#ifndef BO_ORACLE_H
#define BO_ORACLE_H

#include "bo_rules.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BO_INPUT_CHANNELS 120
#define BO_NUM_ACTIONS 4672
#define BO_PLANES_SIZE (BO_INPUT_CHANNELS * 64)

/* utils.RepetitionTracker: Counter over exact transposition keys */
typedef struct {
    bo_key *keys;
    int *counts;
    int n, cap;
} bo_tracker;

int bo_tracker_count(const bo_tracker *t, const bo_key *k);
void bo_tracker_add(bo_tracker *t, const bo_key *k);
void bo_tracker_free(bo_tracker *t);

int bo_move_to_index(bo_move m);                                    /* -1 = ValueError */
int bo_index_to_move(int index, const bo_pos *board, bo_move *out); /* <0 = ValueError */
void bo_encode_board(const bo_pos *hist, int n_hist, const bo_tracker *trk, float *planes);

/* NumPy float32 pairwise summation (np.ndarray.sum on a contiguous f32 array) */
float bo_np_sum_f32(const float *a, long n);

typedef struct {
    int num_simulations;    /* config.NUM_SIMULATIONS  config.py:32 */
    int batch_size;         /* config.MCTS_BATCH_SIZE  config.py:41 */
    double cpuct;           /* config.CPUCT            config.py:33 */
    double widen_coeff;     /* config.WIDEN_COEFF      config.py:40 */
    double dirichlet_alpha; /* config.DIRICHLET_ALPHA  config.py:37 */
    double dirichlet_eps;   /* config.DIRICHLET_EPSILON config.py:39 */
    int max_game_moves;     /* config.MAX_GAME_MOVES   config.py:59 */
} bo_oracle_config;

typedef struct {
    /* the net + softmax: planes[n][120][8][8] -> probs[n][4672], values[n] */
    int (*eval)(void *user, const float *planes, int n, float *probs, float *values);
    /* np.random.dirichlet([alpha]*n_legal) -> out[n_legal] (float64) */
    int (*noise)(void *user, int n_legal, double *out);
    /* self_play.select_move_with_temperature(pi, fullmove_number) -> action index */
    int (*choose)(void *user, const float *pi, int fullmove_number);
    void *user;
} bo_oracle_callbacks;

/* one node of the finished search tree, in creation order (node 0 = root) */
typedef struct {
    int32_t parent;
    int32_t n_visits;
    float q_value;
    float prior;
    bo_move move; /* move from parent */
    int32_t n_children;
    int32_t terminal; /* -1 unknown (never selected as leaf), 0 no, 1 mate, 2 draw */
} bo_oracle_node;

typedef struct {
    int status;             /* 0 ok; 1 = no legal moves at root (reference raises ValueError) */
    bo_move best_move;
    float pi[BO_NUM_ACTIONS];
    int n_nodes;
    bo_oracle_node *nodes;  /* malloc'ed, n_nodes entries; free with bo_oracle_result_free */
    int n_evals;            /* unique positions sent to the evaluator (incl. root) */
    int n_batches;          /* _evaluate_batch calls */
    int n_terminal_sims;    /* simulations absorbed by terminal leaves (mcts.py:235-238) */
    int n_batch_rows;       /* total rows over all batches */
    int max_unique_in_batch;
} bo_oracle_result;

void bo_oracle_result_free(bo_oracle_result *r);

/* run_mcts(root_board, model, history, tracker)  mcts.py:155-280.
 * board: root board incl. its move stack.  hist: boards BEFORE the root (only
 * the last 7 are used, mcts.py:180). */
int bo_oracle_run_mcts(const bo_oracle_config *cfg, const bo_oracle_callbacks *cb, const bo_stack *board,
                       const bo_pos *hist, int n_hist, const bo_tracker *trk, bo_oracle_result *out);

/* one (state, pi, z) record, self_play.py:21 */
typedef struct {
    float state[BO_PLANES_SIZE];
    float pi[BO_NUM_ACTIONS];
    float z;
} bo_record;

typedef struct {
    int status;       /* 0 ok, <0 aborted (reference returns None) */
    int n_records;
    bo_record *records; /* malloc'ed */
    int n_moves;
    bo_move *moves;     /* malloc'ed: the moves played */
    float outcome;      /* utils.get_game_outcome of the final board (or 0.0) */
    int termination;    /* bo_termination_claim_draw of the final board */
    long n_sims;        /* NUM_SIMULATIONS * searches */
    long n_evals;
} bo_game_result;

void bo_game_result_free(bo_game_result *r);

/* run_self_play_game(model, game_id)  self_play.py:84-216; max_plies > 0 stops
 * the game early after that many plies (bench / fixture use; 0 = play out). */
int bo_oracle_self_play(const bo_oracle_config *cfg, const bo_oracle_callbacks *cb, const char *start_fen,
                        int max_plies, bo_game_result *out);

#ifdef __cplusplus
}
#endif
#endif
