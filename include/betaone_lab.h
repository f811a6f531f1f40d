/*
 * include/betaone_lab.h -- measurement and introspection entry points of libbetaone_hip.so that are NOT part of the drop-in boundary.
 *
 * include/betaone_engine.h is what a binding of the reference's hot path (mcts.py / self_play.py / network.py) needs; nothing there
 * depends on this file.  Here: what the parity tests use to look inside a search (bo_debug_tree, bo_debug_fast), what bench.py and
 * scripts/ use to time kernels from inside (bo_nn_tower_forward_timed, bo_debug_stamp, bo_debug_profile, bo_nn_b1_profile,
 * bo_fast_stats, bo_event_pair_overhead), and the synthetic wide-tree PUCT kernel of SURVEY.md section 8d (bo_select_wide: the HBM
 * roofline workload; no search runs it).  Same conventions as betaone_engine.h.  These may change without an ABI bump.
 */
#ifndef BETAONE_LAB_H
#define BETAONE_LAB_H

#include "betaone_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BO_PROF_SLOTS 16             /* uint64 counters per game returned by bo_debug_profile */

/* The tree of the game in `slot` as the search holds it (parity tests compare it node by node with the oracle's and the reference's). */
typedef struct {
    int32_t parent, n_visits, first_child, n_children;
    float q_value, prior;
    int32_t move;      /* from|to<<6|promo<<12 */
    int32_t terminal;  /* -1 never visited as leaf, 0 no, 1 mate, 2 draw */
} bo_node;
int bo_debug_tree(bo_engine *e, int slot, bo_node *out, int32_t cap, int32_t *n_nodes, void *stream);

/* FAST mode, per game [G]: record granules requested by PUCT descents so far (x BO_FAST_GRANULE_BYTES = bytes the select
 * path moved), path nodes updated by the backup (x 16 B; with 12 B x children_scanned + 8 B x levels of bo_engine_status
 * these are the algorithmic bytes of SURVEY.md section 8d), granules in use in the game's arena. */
int bo_fast_stats(bo_engine *e, uint64_t *granules_read, uint64_t *path_nodes, int32_t *arena_granules, int32_t time_select, double *select_ms,
                  int64_t *select_launches, void *stream);
/* time_select: 1 / 0 switches timing of the select + backup kernel (bo_k_fw_select) with HIP events on its launch stream
 * on / off for the following EAGER bo_step calls (not while the stream is being captured), -1 leaves it as it is;
 * select_ms / select_launches return the time and the number of launches accumulated since it was switched on.  Any out
 * pointer may be NULL.  Synchronises. */

/* What a pair of HIP events around ONE kernel launch measures beyond the kernel itself on this device and stream: from the medians of
 * `samples` (1..256) pairs around one and around two empty one-wave kernels (2 * p1 - p2: the second launch's own cost taken out), in
 * milliseconds.  bench.py subtracts it from the event-timed launches of
 * the select + backup kernel (a ~60 us kernel: the pair's own ~6 us is 10 % of it) and checks the result against rocprofv3's
 * per-dispatch durations of the same command (profiles/).  Synchronises `stream`. */
int bo_event_pair_overhead(double *ms_out, int32_t samples, void *stream);

/* FAST mode introspection (tests, profiling): game `slot`'s control block -- ctl_out [ctl_cap >= 16 + 7 * L rounded up to 32]:
 * [0] rows, [1] simulations of the step in flight, [2] live arena, [3] granules in use, [4..8] counters, then from index 16
 * seven arrays of L: row_slot, row_plink, row_nlegal, row_term, row_sim, sim_row, sim_plen (csrc/bo_fastw.h) -- and the paths of
 * the step's simulations, paths_out [L][64] record ids root..leaf.  Either pointer may be NULL.  Synchronises. */
int bo_debug_fast(bo_engine *e, int slot, int32_t *ctl_out, int32_t ctl_cap, int32_t *paths_out, void *stream);

/* Per-phase shader cycles of bo_step (s_memtime), accumulated per game while enabled: cycles_out [G][BO_PROF_SLOTS] (16
 * uint64 per game; size the buffer with the macro) =
 *   [0] apply, [1] select, [2] first visit (move generation + draw rules), [3] terminal backups, [4] leaf encode, [5] flush,
 *   [6] total, [7] game-steps counted, [8] simulation-loop iterations, [9] first visits, [10] terminal-burst calls,
 *   [11] simulations applied inside bursts, [12] terminal simulations on the general path, [13] cycles in burst set-up,
 *   [14] cycles in the burst loop, [15] switches between the two paths a burst holds in registers.
 * enable: 1 = every game-step, N > 1 = only game-steps longer than N cycles, 0 = off (a 0->on switch clears the
 * counters), -1 = only read.  A captured hipGraph keeps the setting it
 * was captured with.  Synchronises when cycles_out != NULL. */
int bo_debug_profile(bo_engine *e, int enable, uint64_t *cycles_out, void *stream);

/* PUCT select (mcts.py:72-118 arithmetic) over caller-provided WIDE trees, the HBM-roofline workload of
 * SURVEY.md section 8d.  blocks_dev: array of 512-byte, 512-byte-aligned child blocks = 32 records
 *   { int32 n; float q; float prior; int32 child_block (-1 = not expanded) }
 * root_block_dev[t] / root_n_dev[t]: root child block and root visit count of tree t; sqrt_lut_dev[n] =
 * f32(sqrt(n + 1e-8)).  out_leaf_dev[t] = block*32 + child of the selected leaf, out_levels_dev[t] = levels
 * descended (x 392 B = algorithmic bytes).  grid_blocks <= 0 picks one 256-thread workgroup per 8 trees.
 * Asynchronous on `stream`. */
int bo_select_wide(const void *blocks_dev, const int32_t *root_block_dev, const int32_t *root_n_dev,
                   const float *sqrt_lut_dev, int n_trees, int max_depth, float cpuct, int grid_blocks,
                   int32_t *out_leaf_dev, int32_t *out_levels_dev, void *stream);

/* (ABI 4) bo_nn_tower_forward with the launch's duration noted by the kernel itself (BO_TOWER_SPLIT_F16): timing_dev = uint64
 * [seq | arrivals | start[4096] | end[4096]], zeroed by the caller; launch k with this buffer leaves (first workgroup's start, last
 * workgroup's end) in slot k % 4096, in ticks of the device's constant-rate clock.  One launch per buffer at a time.  Measurement aid
 * (bench.py's live roofline leg: event pairs cannot sit between the nodes of a captured graph). */
int bo_nn_tower_forward_timed(bo_tower *tower, const float *x_dev, float *y_dev, void *head_a_dev, void *head_b_dev, int batch,
                              void *timing_dev, void *stream);

/* LAB: enqueue a one-thread kernel that appends (tag, wall_clock64()) to `ring_dev` (uint64: [count | tag0, t0 | tag1, t1 | ...], zeroed by
 * the caller, `capacity` entries): the timeline of a stream's phases as the device ran them, also between the nodes of a captured graph. */
int bo_debug_stamp(void *ring_dev, uint64_t tag, uint64_t capacity, void *stream);

/* LAB: per-wave shader-clock sums of a layer's phases {wait, stage, matrix pipe, reduction, epilogue + signal, layers} over the
 * launches between enable = 1 and enable = 0 (which copies [batch * tiles * 4][8] uint64 out, `cap` rows at most).  Not for graphs
 * captured before the switch (the kernel argument is frozen in them). */
int bo_nn_b1_profile(bo_b1 *tower, int enable, uint64_t *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
