/*
 * include/betaone_engine.h -- C ABI of the MI355X self-play rollout engine (libbetaone_hip.so).
 *
 * The reference (kevinh-e/BetaOne) has no FFI layer: its hot path is plain Python
 *   run_mcts(root_board, model, history, tracker)      /root/reference/mcts.py:155-280
 *   _evaluate_batch(nodes, paths, model)               /root/reference/mcts.py:283-295
 *   run_self_play_game(model, game_id)                 /root/reference/self_play.py:84-216
 *   utils.encode_board / move_to_index / index_to_move /root/reference/utils.py:111-365
 * called by main.py:56 and uci.py:63,84.  This header is what a ctypes binding of that path
 * binds instead; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * BO_E_* code and never throws; bo_last_error() gives the text.  "dev" pointers are raw device
 * addresses (tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All kernels are enqueued on `stream`; functions documented as "synchronises" wait for it.
 * One engine = one GPU = G game slots; NN input row g / policy row g / value g belong to slot g.
 */
#ifndef BETAONE_ENGINE_H
#define BETAONE_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an output size, a struct, or the layout a caller has to produce changes (2: bo_debug_profile (betaone_lab.h) returns
 * [G][BO_PROF_SLOTS = 16] counters, the BO_TOWER_WINOGRAD packed-weight K order for 128 filters is winograd_k_order's;
 * 3: fast-mode arenas are allocated in 128-byte granules of 8 records, bo_fast_stats counts granules).  A caller checks
 * bo_abi_version() == BO_ABI_VERSION before anything else (tests/c_abi_smoke.c). */
#define BO_ABI_VERSION 6
#define BO_NUM_ACTIONS 4672          /* config.NUM_ACTIONS, config.py:29 */
#define BO_INPUT_CHANNELS 120        /* config.INPUT_CHANNELS, config.py:28 */
#define BO_ROW_FLOATS (120 * 64)
#define BO_MAX_LEGAL 256
#define BO_RES_CAP 256

enum {
    BO_OK = 0,
    BO_E_ARG = -1,        /* bad argument */
    BO_E_HIP = -2,        /* HIP runtime error */
    BO_E_CONFIG = -3,     /* unsupported configuration (see bo_engine_create) */
    BO_E_FEN = -4,        /* unparsable FEN / UCI move */
    BO_E_STATE = -5       /* call out of order */
};

/* per-game status bits reported by bo_engine_status() */
enum {
    BO_ST_NODE_OVERFLOW = 1, BO_ST_DEPTH_OVERFLOW = 2, BO_ST_NAN_SCORE = 4, BO_ST_PLY_OVERFLOW = 8,
    BO_ST_ILLEGAL_ACTION = 16, BO_ST_UL_OVERFLOW = 32, BO_ST_TRK_OVERFLOW = 64
};

enum { BO_POLICY_NONE = 0, BO_POLICY_LOGITS = 1, BO_POLICY_PROBS = 2 };

/* The constants of config.py that the path reads at call time (config.py:32-41,59). */
typedef struct {
    int32_t n_games;            /* G: game slots resident on this GPU */
    int32_t num_simulations;    /* config.NUM_SIMULATIONS */
    int32_t mcts_batch_size;    /* config.MCTS_BATCH_SIZE */
    int32_t max_plies;          /* capacity of one game's position stack (>= plies + 2) */
    double cpuct;               /* config.CPUCT */
    double widen_coeff;         /* config.WIDEN_COEFF (>= 1.0, int(w*sqrt(batch)) <= 32) */
    double dirichlet_alpha;     /* config.DIRICHLET_ALPHA (only its sign is used on the device) */
    double dirichlet_epsilon;   /* config.DIRICHLET_EPSILON */
    int32_t mode;               /* 0 = the reference's search semantics (bit-exact); 1 = FAST mode (csrc/bo_fastw.h):
                                 * virtual loss, leaves_per_step distinct leaves per game per step, full-width
                                 * expansion -- NOT the reference's semantics */
    int32_t leaves_per_step;    /* FAST mode: L (1..64); NN tensors then have n_games*L rows, row = g*L + r */
    int32_t fast_arena_granules;/* FAST mode: 128-byte granules per game arena (there are two per game); 0 = default, 24 per possible
                                 * expansion of this and the previous search (3 KB; a chess node's run takes ~6).  A run that does
                                 * not fit is refused (BO_ST_NODE_OVERFLOW) and the search goes on with that leaf unexpanded. */
} bo_config;

/* A position as plain data.  bb: pawns, knights, bishops, rooks, queens, kings, white, black. */
typedef struct {
    uint64_t bb[8];
    int32_t turn;             /* 1 white, 0 black */
    uint32_t castling;        /* bit0 K, bit1 Q, bit2 k, bit3 q */
    int32_t ep_square;        /* python-chess Board.ep_square, -1 = None */
    int32_t ep_key;           /* -2: derive (ep square iff an ep capture is legal); else the key's ep (-1 none) */
    int32_t halfmove_clock;
    int32_t fullmove_number;
} bo_position;

typedef struct bo_engine bo_engine;

int bo_abi_version(void);
const char *bo_last_error(void);

int bo_engine_create(const bo_config *cfg, int device, bo_engine **out);
void bo_engine_destroy(bo_engine *e);

/* ---- game set-up ---------------------------------------------------------------------------
 * (Re)start the games in `slots[0..n)`: position `fens[i]` (NULL = standard start, self_play.py:91)
 * followed by the space-separated UCI moves `moves[i]` (NULL = none) -- i.e. a python-chess Board
 * with its move stack, which the draw rules need (mcts.py:36,152).  The repetition tracker holds
 * every position of that stack and the history planes use the <=7 positions before the current
 * one (self_play.py:93-109,182-184).  Also prepares the first search root.  Synchronises. */
int bo_games_reset(bo_engine *e, int n, const int32_t *slots, const char *const *fens, const char *const *moves,
                   void *stream);

/* Same, but with the caller's own history boards and tracker contents (uci.py:62-63 passes
 * whatever it accumulated): hist[i*7 .. i*7+n_hist[i]) boards BEFORE the root (oldest first),
 * tracker keys trk[trk_off[i] .. trk_off[i+1]) with their counts.  Synchronises. */
int bo_games_reset_ex(bo_engine *e, int n, const int32_t *slots, const char *const *fens, const char *const *moves,
                      const bo_position *hist, const int32_t *n_hist, const bo_position *trk,
                      const int32_t *trk_counts, const int32_t *trk_off, void *stream);

/* Root facts the host needs before a search: number of legal moves (np.random.dirichlet needs it,
 * mcts.py:191-192) and is_game_over(claim_draw=True) of the current position (0 no, 1 side to move
 * is checkmated, 2 draw; self_play.py:101-102).  Synchronises.  Arrays are [G]. */
int bo_root_info(bo_engine *e, int32_t *n_legal, int32_t *terminal, int32_t *ply, void *stream);

/* ---- one search per game, all games in lock step ------------------------------------------------
 * bo_search_begin: start run_mcts for every slot with go[g] != 0.  noise[g*256 + i] is the
 * Dirichlet sample of the i-th legal move of game g in python-chess order (mcts.py:192-198);
 * may be NULL when dirichlet_alpha <= 0.  Writes planes 0..97 of NN input row g (nn_in_dev,
 * float32 [G,120,8,8], NCHW contiguous). */
int bo_search_begin(bo_engine *e, const int32_t *go, const double *noise, float *nn_in_dev, void *stream);

/* bo_step: consume the net's output for the rows requested by the previous step (policy_dev
 * float32 [G,4672] logits or softmax probabilities per `policy_kind`, value_dev float32 [G]),
 * run select / terminal backups / expand+backup flushes until each game needs its next
 * evaluation, and write that leaf's planes into nn_in_dev row g.  First call of a search:
 * policy_kind = BO_POLICY_NONE.  Asynchronous on `stream`. */
int bo_step(bo_engine *e, const float *policy_dev, const float *value_dev, int policy_kind, float *nn_in_dev,
            void *stream);
/* (ABI 6) bo_step with the TAIL of the evaluate stage inside the step kernel: behind bo_nn_heads(flags & 4) -- which stops after the
 * logits and the partial sums of value_fc1 -- the wave that consumes a game's row does what the stage's rows kernel would have done
 * with it: the softmax of mcts.py:185,287 (at the indices it needs) and value = tanh(value_fc2(relu(value_fc1 + bias)))
 * (network.py:195-197).  Same float32 operations in the same order as bo_nn_heads' own second launch: the search sees the same
 * bits either way (tests/test_baseline_configs_gpu.py), one launch and one pass over the rows less per evaluation.
 * logits_dev [G,4672]; vpart_dev = bo_nn_heads' scratch_dev [16][rows][256] (rows = that call's batch >= G, row g <-> game g);
 * b1_dev [256], w2_dev [256], b2_dev [1] = value_fc1.bias, value_fc2.weight, value_fc2.bias.  Reference-semantics engines only. */
int bo_step_heads(bo_engine *e, const float *logits_dev, const float *vpart_dev, const float *b1_dev, const float *w2_dev,
                  const float *b2_dev, int rows, float *nn_in_dev, void *stream);

/* How many searches are still running / how many rows were requested by the last step.
 * requested_mask (optional, [G] int32) marks the rows the net must evaluate.  Synchronises. */
int bo_search_poll(bo_engine *e, int32_t *n_running, int32_t *n_requested, int32_t *requested_mask, void *stream);

/* Interrupt running searches between two steps (the reference can only stop between whole
 * searches, uci.py:73): for every game with stop_mask[g] != 0 (NULL = all) whose search is running, the pending rows are
 * flushed exactly like the tail batch of mcts.py:256-257, an evaluation still outstanding is dropped, and the search is
 * marked finished -- bo_search_result then returns what the reference returns for NUM_SIMULATIONS = sims_done[g]
 * (optional out, [G] int32: simulations completed per game).  Reference-semantics engines only.  Synchronises. */
int bo_search_stop(bo_engine *e, const int32_t *stop_mask, int32_t *sims_done, void *stream);

/* Result of the finished searches (mcts.py:259-280): sparse pi (res_n[g] entries of
 * (action index, probability) at [g*BO_RES_CAP ..]), best move as action index (-1: no legal
 * move, the reference raises ValueError) and as from|to<<6|promo<<12, total root-child visits.
 * Synchronises. */
int bo_search_result(bo_engine *e, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx,
                     int32_t *best_move, int32_t *total_visits, void *stream);

/* self_play.py:125-184: play action[g] (an index into the 4672 actions; -2 = play the search's
 * best move; -1 = leave the game alone) with the reference's decode-error / illegal-move
 * fallbacks, push it on the game's stack and tracker, prepare the next root.  Asynchronous: the kernel reads the actions from a
 * two-deep pinned ring of the engine's when it runs; a call whose half of the ring is still unread (two calls back-to-back without
 * the device catching up) waits for that kernel first, so any number of calls may be enqueued. */
int bo_play(bo_engine *e, const int32_t *action, void *stream);

/* ---- per-move host work, natively -----------------------------------------------------------------
 * The reference draws np.random.dirichlet (mcts.py:192) and np.random.choice (self_play.py:73) once per move.
 * With one legacy MT19937 stream per game slot kept inside the engine (bit-compatible with
 * numpy.random.RandomState(seed): same seeding, same legacy gamma/dirichlet algorithm, same draws), the per-move
 * host work of thousands of games is two calls:
 *   bo_selfplay_sample : bo_search_result + select_move_with_temperature (self_play.py:59-80) for every active
 *                        game; action_out[g] = sampled action index, -1 inactive, -3 "pi has more than two
 *                        non-zero entries (or temperature 0): sample with the dense NumPy mirror" (the caller
 *                        may move the stream out and back with bo_rng_state).  Synchronises.
 *   bo_selfplay_begin  : bo_root_info + Dirichlet noise for every game with want[g] != 0 whose root is not
 *                        terminal + bo_search_begin + the first bo_step.  Synchronises (root info). */
int bo_rng_seed(bo_engine *e, int slot, uint32_t seed);
int bo_rng_state(bo_engine *e, int slot, int set, uint32_t *key624, int32_t *pos, int32_t *has_gauss, double *gauss);
int bo_selfplay_sample(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold,
                       double t_initial, double t_final, int32_t *res_n, int32_t *res_idx, float *res_val,
                       int32_t *best_idx, int32_t *action_out, void *stream);
int bo_selfplay_begin(bo_engine *e, const int32_t *want, float *nn_in_dev, int32_t *n_legal_out, int32_t *terminal_out,
                      int32_t *go_out, void *stream);

/* bo_selfplay_sample + bo_play + bo_selfplay_begin(want_next) in one call (one host round trip per ply).  *completed = 0
 * if some game's pi was too dense for the native sampler (action -3): nothing was played, the caller samples that game
 * with the dense NumPy mirror, then calls bo_play and bo_selfplay_begin itself. */
int bo_selfplay_turn(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold, double t_initial,
                     double t_final, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx, int32_t *action_out,
                     const int32_t *want_next, float *nn_in_dev, int32_t *n_legal_out, int32_t *terminal_out, int32_t *go_out,
                     int32_t defer_noise, int32_t *completed, void *stream);
/* defer_noise is a bit set.  Bit 1 (value 2): first ask whether every search has finished (bo_search_poll); if one is still
 * running nothing is done and *completed = -1 -- the caller issues another evaluate + step and calls again.
 * Bit 0, defer_noise = 1: the Dirichlet draws of the new roots (mcts.py:190-201) and their upload are left to bo_selfplay_noise,
 * to be called after the root evaluations' network forward has been enqueued on `stream` (the host work overlaps it) and
 * before the bo_step that consumes those evaluations.  Per game the RNG stream order is the same either way.
 * Bit 2 (value 4, with bit 0): the begin does not wait for the device either -- the kernel itself starts the search of every
 * wanted game whose new root is not terminal (mcts.py:160-162), *completed = 2, and n_legal_out / terminal_out / go_out are
 * NOT written: bo_selfplay_begun returns them (as bo_selfplay_begin would have) once the caller has enqueued the first
 * evaluation; it waits for the copy behind the begin kernels only.  Order: bo_selfplay_turn(4) -> enqueue the network
 * forward -> bo_selfplay_begun -> bo_selfplay_noise -> bo_step.  (Reference-semantics engines.)
 * Bit 3 (value 8, with bit 1): the result block and the searches' state have been enqueued behind the searches already
 * (bo_search_result_prefetch) and no step has been issued since: the call then only waits for `stream` and reads both from pinned
 * memory -- a cohort whose stream is idle is turned without a device round trip. */
int bo_selfplay_begun(bo_engine *e, int32_t *n_legal_out, int32_t *terminal_out, int32_t *go_out);
/* (ABI 4) Enqueue the result kernel and the copy of [result block | searches' state] to pinned host memory on `stream`, behind the
 * searches' last expected step, and return without waiting (see bo_selfplay_turn, bit 3).  Replaying a CAPTURED step afterwards makes the
 * block stale without the library knowing: the caller prefetches again (bo_step itself invalidates it). */
int bo_search_result_prefetch(bo_engine *e, void *stream);
int bo_selfplay_noise(bo_engine *e, void *stream);

/* (ABI 5) The turn of a ply ON THE DEVICE: bo_selfplay_turn's work -- result, select_move_with_temperature (self_play.py:59-80), the
 * played move (self_play.py:125-184) and the begin of the next searches (mcts.py:160-162) -- enqueued on `stream` BEHIND the searches'
 * last expected bo_step, so that the device goes from a ply's last tree step straight into the next ply's root evaluation; the host
 * is not waited for.  What stays on the host is every random draw and every libm call: per game with active[g] != 0 this call draws the
 * uniform np.random.choice would draw for the move (self_play.py:73) from the slot's stream NOW (the stream order per game is unchanged:
 * Dirichlet of this search, choice of this move, Dirichlet of the next search) and hands it to the kernel; apply_temperature's
 * p ** (1 / T) comes from a table built here with the host's pow for every visit count 0..NUM_SIMULATIONS.  The rest of the sampling is
 * IEEE arithmetic restated on the device for a pi of <= 2 non-zero entries (the reference's root keeps <= 2 children).
 * Needs a reference-semantics engine with int(WIDEN_COEFF) == 1, t_initial == 1, t_final > 0 (BO_E_CONFIG otherwise: use
 * bo_selfplay_turn).  redo != 0: enqueue the same turn again with the draws already made (after *completed == -1 below).
 * Order per ply: [bo_selfplay_noise -> bo_step x n] -> bo_selfplay_autoturn -> enqueue the next root evaluation's network forward ->
 * bo_selfplay_autoturn_collect -> bo_selfplay_noise -> bo_step ...  Asynchronous. */
int bo_selfplay_autoturn(bo_engine *e, const int32_t *active, const int32_t *move_number, int32_t threshold, double t_initial,
                         double t_final, const int32_t *want_next, float *nn_in_dev, int32_t redo, void *stream);
/* *ready_out = 1 once the device has passed the turn's outputs (bo_selfplay_autoturn_collect will not wait), else 0.  Never blocks. */
int bo_selfplay_autoturn_ready(bo_engine *e, int32_t *ready_out);
/* Wait for the turn's outputs (not for work enqueued behind them) and return them: the sparse pi of every game that searched (res_n,
 * res_idx / res_val rows of BO_RES_CAP like bo_search_result; <= 2 entries), best_idx (may be NULL), the action played (-1: none), and --
 * as bo_selfplay_begin reports them -- the new roots' legal-move counts, terminal codes and go flags (any may be NULL).
 * *completed = 1: done; -1: some search was still running when the turn came up -- NOTHING was played or begun: issue one more
 * evaluation + bo_step, then bo_selfplay_autoturn(redo = 1).  A result the device sampler does not cover is BO_E_STATE. */
int bo_selfplay_autoturn_collect(bo_engine *e, int32_t *res_n, int32_t *res_idx, float *res_val, int32_t *best_idx, int32_t *action_out,
                                 int32_t *n_legal_out, int32_t *terminal_out, int32_t *go_out, int32_t *completed);

/* ---- records ------------------------------------------------------------------------------------
 * The game in `slot` as plain data: its positions[0..n_plies] and moves[0..n_plies). */
int bo_game_export(bo_engine *e, int slot, bo_position *positions, int32_t *moves, int32_t cap, int32_t *n_plies,
                   void *stream);
/* Training encodings of plies [first, first+n) of the game in `slot` with the CURRENT (end of
 * game) tracker, float32 [n,120,8,8] into out_dev (self_play.py:200-208).  Asynchronous. */
int bo_game_encode(bo_engine *e, int slot, int first, int n, float *out_dev, void *stream);

/* The same encodings from a compact record on any rank (betaone_amd/records.py wire format): positions[0..n_positions)
 * as returned by bo_game_export (ep_key filled), plies [first, first+n) into out_dev.  Needs no engine.  Synchronises. */
int bo_records_encode(int n_positions, const bo_position *positions, int first, int n, float *out_dev, void *stream);

/* ---- FAST mode options ---------------------------------------------------------------------------- */
/* FAST mode (cfg.mode = 1; NOT the reference's semantics).  Every argument: -1 leaves the
 * setting as it is.  tree_reuse != 0 (default) keeps the played child's subtree as the next search's tree -- the reference
 * rebuilds the tree every move (mcts.py:176).  games_per_halfwave (2 or 4, default 2): games the select + backup kernel
 * interleaves per half-wavefront in the half-wave forms (leaves_per_step > 8 caps it at 2, > 16 at 1).  select_flags (default 16):
 * bit 0 non-temporal loads of the child runs, bit 1 the root's run stays in registers for all descents of a step, bit 2 the
 * half-wave kernel is built for one more wavefront per SIMD (register spills), bit 3 (leaves_per_step == 4 only) the
 * one-lane-per-game form, bit 4 (leaves_per_step <= 8) eight lanes per game = eight games per wave-instruction.
 * All variants compute the same trees (tests/test_engine_gpu.py); the defaults are the fastest measured on an MI355X. */
int bo_fast_options(bo_engine *e, int32_t tree_reuse, int32_t games_per_halfwave, int32_t select_flags);
#define BO_FAST_GRANULE_BYTES 128    /* a fast-mode arena is allocated in granules of 8 16-byte records */
/* per game [G]: status bits, NN evaluations, flushes, terminal simulations, tree levels descended,
 * children scanned by the PUCT select (the last two give the select kernel's algorithmic bytes). */
int bo_engine_status(bo_engine *e, int32_t *status, int32_t *evals, int32_t *flushes, int32_t *term_sims,
                     int32_t *levels, int32_t *children_scanned, void *stream);


/* ---- stand-alone kernels -------------------------------------------------------------------------
 * Legal moves (python-chess order) of n raw positions: moves_out [n,256] int32, n_out [n], check_out [n]. */
int bo_movegen_batch(bo_engine *e, int n, const bo_position *pos, int32_t *moves_out, int32_t *n_out,
                     int32_t *check_out, void *stream);


/* ---- fused epilogues of the evaluate stage (network.py:64-118 with BatchNorm folded), NCHW float32, 8x8 ----------
 * x = relu(x + bias[c] (+ residual)) in place; residual_dev may be NULL.  Asynchronous on `stream`. */
int bo_nn_bias_act(float *x_dev, const float *bias_dev, const float *residual_dev, int batch, int channels, void *stream);
/* x = relu((x + bias[c]) * sigmoid(W2 relu(W1 mean_hw(x + bias))) + residual) in place: the SE residual block's tail
 * (network.py:33-45,100-118).  w1 [hidden][channels], w2 [channels][hidden].  Asynchronous on `stream`. */
int bo_nn_se_residual(float *x_dev, const float *bias_dev, const float *w1_dev, const float *w2_dev,
                      const float *residual_dev, int batch, int channels, int hidden, void *stream);

/* 3x3 convolution (padding 1) over 8x8 boards on the fp32 matrix cores with the epilogue fused, NCHW float32:
 *   mode 0: y = conv(x) + bias      1: y = relu(conv(x) + bias)      2: y = relu(conv(x) + bias + residual)
 * wpacked_dev: the [c_out][c_in][3][3] weights re-ordered as [tap 9][c_in/8][c_out][2][4] with element
 * (tap, t4, oc, k, e) = W[oc][8*t4 + 2*e + k][tap] (betaone_amd/fused_net.py:pack_conv_weight).
 * Supported (c_in, c_out): (120 | C, C) for C in {64, 128, 256}.  Asynchronous on `stream`. */
/* Small-batch form of bo_nn_se_residual (uci.py's single-position searches): a board's layer in channels/16 workgroups;
 * x_dev is only read, the result replaces residual_inout_dev: r = relu((x + bias[c]) * gate[b,c] + r).  channels a multiple
 * of 16 (<= 256), hidden <= 16. */
int bo_nn_se_residual_small(const float *x_dev, const float *bias_dev, const float *w1_dev, const float *w2_dev,
                            float *residual_inout_dev, int batch, int channels, int hidden, void *stream);
/* Everything behind the tower's head convolutions in two launches (csrc/bo_heads.h; network.py:186-197 + the softmax of
 * mcts.py:185,287): policy_out = softmax(policy_fc(p)) (flags bit 0 clear: the logits), value_out = tanh(value_fc2(relu(value_fc1(v)))).
 * p [batch,128], v [batch,2048], policy_fc weight [4672,128] + bias, value_fc1 weight [256,2048] + bias, value_fc2 weight
 * [256] + bias [1], all float32 row-major on the device; policy_out [batch,4672], value_out [batch].  scratch_dev:
 * 4096 * batch floats (value_fc1's partial sums, no initial contents needed).  Any batch up to 65536.
 * flags: bit 0 = softmax; bit 1 (value 2) = p and v are float16 (the head planes of the BO_TOWER_DIRECT_F16 tower): they are
 * widened on load, weights, accumulation and outputs stay float32; bit 2 (value 4, ABI 6; excludes bit 0) = stop after the first
 * launch: policy_out holds the logits, scratch_dev the partial sums, value_out_dev is not written (may be NULL) -- bo_step_heads
 * finishes both where they are consumed. */
int bo_nn_heads(const void *p_dev, const void *v_dev, const float *wp_dev, const float *bp_dev, const float *w1_dev,
                const float *b1_dev, const float *w2_dev, const float *b2_dev, float *policy_out_dev, float *value_out_dev,
                float *scratch_dev, int batch, int flags, void *stream);
/* (ABI 4) The same behind the fp16 tower at any number of rows (fast mode: 4 096 .. 131 072 rows per evaluation; network.py:186-197 under
 * torch.autocast, softmax of mcts.py:287 in float32): p [batch,128], v [batch,2048], policy_fc weight [4672,128] and value_fc1 weight
 * [256,2048] float16 row-major; biases, value_fc2 weight [256] + bias float32; products on the fp16 matrix pipe with float32
 * accumulation, logits never rounded to float16.  The probabilities are written once: a first launch leaves every board's (max, sum of
 * exp) per output range in scratch_dev (20 * batch floats, no initial contents needed; may be NULL without softmax), a second
 * computes every logits tile again and stores exp(x - max) / sum; a third launch is the whole value head.  flags bit 0 = softmax.
 * 1 <= batch <= 4194304. */
int bo_nn_heads_f16(const void *p_dev, const void *v_dev, const void *wp_f16_dev, const float *bp_dev, const void *w1_f16_dev,
                    const float *b1_dev, const float *w2_dev, const float *b2_dev, float *policy_out_dev, float *value_out_dev,
                    float *scratch_dev, int batch, int flags, void *stream);
int bo_nn_conv3x3(const float *x_dev, const float *wpacked_dev, const float *bias_dev, const float *residual_dev,
                  float *y_dev, int batch, int c_in, int c_out, int mode, void *stream);

/* Small-batch form of bo_nn_conv3x3 (uci.py's single-position analysis): (c_out/16) x 4 workgroups per board, K split
 * over the four waves of a workgroup.  wpacked_dev: [c_out/16][tap 9][c_in/16][64][4] with element (ot, tap, g, lane, e) =
 * W[16*ot + (lane & 15)][16*g + 4*e + (lane >> 4)][tap]; c_in in {64, 128, 256} is the (zero-padded) channel count of
 * the weights, c_in_x <= c_in the channel count of x (120 for the input conv with c_in = 128).  Same modes. */
int bo_nn_conv3x3_small(const float *x_dev, const float *wpacked_dev, const float *bias_dev, const float *residual_dev, float *y_dev,
                        int batch, int c_in, int c_in_x, int c_out, int mode, void *stream);

/* The whole residual tower (input conv + N residual blocks, /root/reference/network.py:48-118,167-190, BatchNorm
 * folded) as ONE persistent kernel that keeps each board's activations in LDS.  channels in {64, 128}.
 * Layer l: kind 0 = input conv 120 -> C with ReLU (first layer only; weights zero-padded to 128 input channels),
 * 1 = first conv of a block with ReLU, 2 = second conv + skip + ReLU, 3 = second conv + SE gate + skip + ReLU
 * (W1 [hidden][C], W2 [C][hidden], no biases, hidden <= 16); `last` = 1 on the final layer only.
 * algo BO_TOWER_DIRECT (csrc/bo_tower.h): implicit GEMM; t4 = c_in/8; weights per layer [tap 9][t4][C][2][4] as for
 *   bo_nn_conv3x3.
 * algo BO_TOWER_WINOGRAD (csrc/bo_tower_wg.h): F(2x2,3x3); t4 = c_in/4 K-steps; weights per layer
 *   [t4][C/16][4][64][4] with element (step, ob, pq, lane, e) = (G g G^T)[4*pq + e] of filter
 *   g = W[16*ob + (lane & 15)][channel(step, lane >> 4)], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], position = 4*row + col;
 *   channel(step, k) = 4*step + k for C = 64 and, for C = 128 (every layer has 128 input channels, the first one padded),
 *   with step = 4*c + sl: 16*(4*(c & 1) + sl) + 4*(c >> 1) + k -- the K order in which a wave of the kernel only ever
 *   transforms input channels that it produced itself (csrc/bo_tower_wg.h: OWN).  bias_off must be a multiple of 4.
 * algo BO_TOWER_DIRECT_F16 (csrc/bo_tower_h.h): fp16 weights and activations, fp32 accumulation, two boards per
 *   workgroup; channels in {128, 256}; t4 = 9*c_in/16 K-steps; `weights` holds fp16 data (n_weights still counts
 *   4-byte units): per layer [t4][C/32][64][8] with element (step, mt, lane, i) = W[32*mt + (lane & 31)]
 *   [16*(step % (c_in/16)) + 8*(lane >> 5) + i][tap = step / (c_in/16)] at 16-byte offset w_off4.  Needs `head`: its
 *   weights [ceil(channels/32)][C/16][64][8] fp16 with element (mt, st, lane, i) = Wh[32*mt + (lane & 31)][16*st +
 *   8*(lane >> 5) + i] at 16-byte offset w_off in `weights`; head outputs are fp16; y_dev is unused.
 * algo BO_TOWER_SPLIT_F16 (csrc/bo_tower_s.h): float32 in and out, computed on the fp16 matrix pipe -- every float32 weight
 *   and activation is a (hi, lo) pair of fp16 values (hi = RN16(v), lo = RN16(v - hi)), every product three fp16 MFMAs with
 *   float32 accumulation (relative product error 2^-22); one board per workgroup; channels in {128, 256}; layout as
 *   BO_TOWER_DIRECT_F16 with every fragment doubled: per layer [t4][C/32][2 = hi, lo][64][8] fp16 of s*W, s a power of two
 *   chosen by the caller (largest |s*W| below 2^15), and 1/s as ONE MORE float behind the layer's bias (params[bias_off + C];
 *   bias_off a multiple of 4).
 *   `head` is optional: weights [ceil(channels/32)][C/16][2][64][8] at 16-byte offset w_off in `weights`, bias [channels]
 *   followed by the inverse scale in params; head outputs are float32.  y_dev (optional) receives the tower output.
 * algo BO_TOWER_SPLIT_F16_T16 (ABI 5, csrc/bo_tower_s16.h; 128 filters): BO_TOWER_SPLIT_F16 with the products issued as 16x16x32 tiles (the chip
 *   holds a higher clock under them: profiles/r05_tower_bound.md); everything as for BO_TOWER_SPLIT_F16 except the per-layer fragment order
 *   [tap 9][c_in/32][C/16][2 = hi, lo][64][8]: element (tap, g, ot, hl, lane, i) = s*W[16*ot + (lane & 15)][32*g + 8*(lane >> 4) + i][tap]
 *   (t4 still counts K-steps of 16 channels: 9*c_in/16).
 * algo BO_TOWER_DIRECT_F16_T16 (ABI 5, csrc/bo_tower_h16.h; 128 or 256 filters): BO_TOWER_DIRECT_F16 as 16x16x32 tiles; per-layer fragment order
 *   [tap 9][c_in/32][C/16][64][8]: element (tap, g, ot, lane, i) = W[16*ot + (lane & 15)][32*g + 8*(lane >> 4) + i][tap]; head weights unchanged.
 * weights: float32 at float4 offset w_off4; params: float32 biases and SE matrices at float offsets.
 * head (optional, BO_TOWER_WINOGRAD only): the policy and value 1x1 convolutions + ReLU (network.py:101-113,191-195)
 *   fused behind the tower: bias [channels] and weights in params, the weights packed [ceil(channels/16)][C/16][64][4]
 *   with element (mb, g, lane, e) = Wh[16*mb + (lane & 15)][16*g + 4*e + (lane >> 4)] (0 beyond the last channel) at
 *   a float offset w_off that is a multiple of 4; output channels [0, split) go to
 *   head_a_dev [batch][split][64], the rest to head_b_dev [batch][channels - split][64]; y_dev may then be NULL.
 * bo_nn_tower_create validates every offset and copies the three HOST arrays to `device`;
 * bo_nn_tower_forward(x_dev [batch,120,8,8] -> y_dev [batch,C,8,8], NCHW float32) is asynchronous on `stream`.
 * bo_nn_value_tail: out[b] = tanh(w . h[b] + bias[0]) (value_fc2 + tanh, network.py:116-118,197). */
enum { BO_TOWER_DIRECT = 0, BO_TOWER_WINOGRAD = 1, BO_TOWER_DIRECT_F16 = 2, BO_TOWER_SPLIT_F16 = 3, BO_TOWER_SPLIT_F16_T16 = 4, BO_TOWER_DIRECT_F16_T16 = 5 };
typedef struct bo_tower_layer_desc {
    int32_t w_off4, t4, bias_off, kind, se_w1_off, se_w2_off, hidden, last;
} bo_tower_layer_desc;
typedef struct bo_tower_head_desc {
    int32_t channels, split, w_off, b_off;
} bo_tower_head_desc;
typedef struct bo_tower_s bo_tower;
int bo_nn_tower_create(const bo_tower_layer_desc *layers, int n_layers, const float *weights, int64_t n_weights,
                       const float *params, int64_t n_params, int channels, int algo, const bo_tower_head_desc *head, int device,
                       bo_tower **out);
int bo_nn_tower_forward(bo_tower *tower, const float *x_dev, float *y_dev, void *head_a_dev, void *head_b_dev, int batch, void *stream);
int bo_device_wall_clock_khz(int device, int32_t *khz_out);   /* rate of that clock (hipDeviceAttributeWallClockRate) */
/* (ABI 4) A HIP stream confined to a set of compute units (hipExtStreamCreateWithCUMask): bit i of mask_words = CU i of `device`.
 * CohortRollout gives every cohort such a stream with a disjoint set: the stream has a hardware queue of its own (pool streams share
 * queues: four cohorts then wait for each other's launches) and its kernels stay off the other cohorts' CUs (the reference's counterpart
 * is one OS process per game batch, main.py:160-175).  The handle is a
 * hipStream_t for every `stream` argument of this header and for torch.cuda.ExternalStream. */
int bo_stream_create_cu_mask(int device, const uint32_t *mask_words, int n_words, void **stream_out);
int bo_stream_destroy(void *stream);
int bo_nn_value_tail(const float *h_dev, const float *w_dev, const float *bias_dev, float *out_dev, int batch, int hidden, void *stream);
void bo_nn_tower_destroy(bo_tower *tower);
/* BO_TOWER_SPLIT_F16 carries every activation as a pair of fp16 numbers: a value beyond +-65504 is saturated and the forward's result
 * is wrong.  *overflow_out = 1 if that happened in any forward since the last call (the flag is cleared), 0 otherwise (always 0 for
 * the other algorithms).  Synchronises `stream`.  A net that trips it needs the fp32-pipe tower (BETAONE_F32_TOWER=fp32). */
int bo_nn_tower_status(bo_tower *tower, int32_t *overflow_out, void *stream);
/* (ABI 4) Device address of that status word, for bo_engine_watch: the self-play loop then checks it with every ply's result block
 * instead of only when finished games are handed over (the reference has no counterpart: its float32 net cannot saturate). */
int bo_nn_tower_word(bo_tower *tower, void **dev_word_out);
/* (ABI 4) Let the engine's result kernels copy `*dev_word` (any int32 device word, NULL: none) behind the result block, so that
 * bo_search_result / bo_selfplay_turn bring it to the host in the round trip they make anyway; bo_engine_watch_seen returns the OR
 * of the values seen since the last call with clear != 0.  Nothing is enqueued and nothing waits in either call. */
int bo_engine_watch(bo_engine *engine, int32_t *dev_word);
/* (ABI 5) The same for n_words (1 or 2) consecutive words: the copy is word 0 | (word 1 != 0 ? 0x10000 : 0).  Two words are what
 * bo_nn_b1_word returns -- the one-launch tower's [hand-off timeout code | saturation flag] -- so uci.py's searches and small self-play
 * batches, which that tower evaluates, stop on an invalid evaluation instead of using it. */
int bo_engine_watch_words(bo_engine *engine, int32_t *dev_words, int32_t n_words);
int bo_engine_watch_seen(bo_engine *engine, int32_t *seen_out, int32_t clear);


/* ---- (ABI 4) GPU-resident replay buffer: csrc/bo_replay.h ----------------------------------------------------------------------
 * Replaces, on the training side of the path's hand-over, train.load_recent_data + ChessDataset (/root/reference/train.py:179-219): the
 * finished games stay in HBM as compact records (position 80 B, end-of-game repetition count, sparse pi, z: ~110 B per ply) and a
 * batch of the triples ChessDataset.__getitem__ yields -- state float32 [120,8,8], dense pi [4672], z -- is expanded on the device for
 * the loop that consumes it (train_network, train.py:252-262).  capacity in POSITION slots (a game of n records takes n + 1); the
 * oldest games are evicted when the ring comes round (the reference keeps the most recent iterations, train.py:190-193). */
typedef struct bo_replay_s bo_replay;
int bo_replay_create(int64_t capacity_positions, int pi_width, int device, bo_replay **out);
/* positions[0 .. n_records] (position i before move i; the last one final), pi of record i = entries pi_ptr[i] .. pi_ptr[i+1] (at most
 * pi_width), z[i] as self_play.py:202 stores it.  *evicted_records (may be NULL): records of the games that had to go. */
int bo_replay_add_game(bo_replay *rb, int32_t game_id, const bo_position *positions, int32_t n_records, const int32_t *pi_ptr,
                       const int32_t *pi_idx, const float *pi_val, const float *z, int64_t *evicted_records, void *stream);
int bo_replay_size(bo_replay *rb, int64_t *n_records, int64_t *n_games);
/* record_index[i] in [0, records): resident records, oldest game first.  states [n,120,8,8], pi [n,4672], z [n] (device, float32). */
int bo_replay_sample(bo_replay *rb, int32_t n, const int64_t *record_index, float *states_dev, float *pi_dev, float *z_dev, void *stream);
void bo_replay_destroy(bo_replay *rb);

/* ---- (ABI 4) the residual tower of ONE board (a few boards) as ONE launch spread over the chip: csrc/bo_tower_b1.h ----------------
 * Replaces, for uci.py's single-position searches (/root/reference/uci.py:60-93 -> mcts.py:183-185: PolicyValueNet.forward at
 * batch 1, /root/reference/network.py:167-185), the per-layer launches of bo_nn_conv3x3_small / bo_nn_se_residual_small.
 * Layer l = 0 is the input convolution (weights packed for 128 input channels, c_in_x = 120 present), then (conv1, conv2) per
 * residual block: mode 0 = relu(conv + bias), 1 = relu(conv + bias + block input), 2 = relu((conv + bias) * SE gate + block input)
 * with se_w1 [hidden][C], se_w2 [C][hidden] (network.py:33-45), hidden <= 16.  Weights: the layout of bo_nn_conv3x3_small.
 * All pointers are device addresses that must stay valid for the handle's life.  (filters / 16) * 4 * batch <= 256. */
typedef struct bo_b1_layer_desc {
    const void *weights_dev;
    const float *bias_dev;
    const float *se_w1_dev, *se_w2_dev;
    int32_t c_in, c_in_x, mode, se_hidden;
    const void *weights_split_dev;   /* NULL, or (hi, lo) fp16 pairs of inv_scale^-1 * W: the tiles then multiply on the fp16 matrix pipe
                                        (three MFMAs per product, float32 accumulation: the precision of BO_TOWER_SPLIT_F16); for every
                                        layer or for none.  Layout [C/16][tap 9][c_in/16][64 lanes][hi x4 | lo x4]. */
    float inv_scale;                 /* 1 / (the power of two the split weights were scaled by) */
    int32_t reserved;
} bo_b1_layer_desc;
typedef struct bo_b1_s bo_b1;
int bo_nn_b1_create(const bo_b1_layer_desc *layers, int n_layers, int channels, int max_batch, int device, bo_b1 **out);
/* x [batch,120,8,8] -> y [batch,channels,8,8] float32 on `stream`; capturable; one launch of a handle in flight at a time. */
int bo_nn_b1_forward(bo_b1 *tower, const float *x_dev, float *y_dev, int batch, void *stream);
/* *code_out = 0; 1 + the phase of a hand-off wait that gave up in the last launch (bounded spins: the kernel always ends; its output
 * is invalid then); or -1: an activation left the fp16 range with split weights (saturated: the output is wrong, use a handle without
 * split weights).  Synchronises `stream`. */
int bo_nn_b1_status(bo_b1 *tower, int32_t *code_out, void *stream);
/* (ABI 5) Device address of the two status words bo_nn_b1_status reads ([timeout code | saturation flag]; sticky until that call). */
int bo_nn_b1_word(bo_b1 *tower, void **dev_words_out);
void bo_nn_b1_destroy(bo_b1 *tower);

#ifdef __cplusplus
}
#endif
#endif
