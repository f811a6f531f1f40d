/* tests/c_abi_smoke.c -- a plain-C consumer of include/betaone_engine.h (no Python, no torch): creates an engine,
 * sets up two games, reads root facts, runs the move generator kernel, and drives one search with a constant
 * policy/value supplied from plain hipMalloc'ed buffers.  Built and run by tests/test_c_abi_gpu.py on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "betaone_engine.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, bo_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(void) {
    if (bo_abi_version() != BO_ABI_VERSION) return 2;
    bo_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.n_games = 2; cfg.num_simulations = 200; cfg.mcts_batch_size = 96; cfg.max_plies = 64;
    cfg.cpuct = 1.0; cfg.widen_coeff = 1.5; cfg.dirichlet_alpha = 0.0; cfg.dirichlet_epsilon = 0.25;
    cfg.mode = 0; cfg.leaves_per_step = 1;
    bo_engine *e = NULL;
    CHECK(bo_engine_create(&cfg, 0, &e));
    int32_t slots[2] = {0, 1};
    const char *fens[2] = {NULL, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"};
    const char *moves[2] = {"e2e4 e7e5", NULL};
    CHECK(bo_games_reset(e, 2, slots, fens, moves, NULL));
    int32_t nl[2], term[2], ply[2];
    CHECK(bo_root_info(e, nl, term, ply, NULL));
    printf("n_legal %d %d terminal %d %d ply %d %d\n", nl[0], nl[1], term[0], term[1], ply[0], ply[1]);
    if (nl[0] != 29 || nl[1] != 48 || term[0] || term[1] || ply[0] != 2 || ply[1] != 0) return 3;

    float *nn_in, *policy, *value;
    HIP(hipMalloc((void **)&nn_in, 2 * BO_ROW_FLOATS * sizeof(float)));
    HIP(hipMalloc((void **)&policy, 2 * BO_NUM_ACTIONS * sizeof(float)));
    HIP(hipMalloc((void **)&value, 2 * sizeof(float)));
    HIP(hipMemset(policy, 0, 2 * BO_NUM_ACTIONS * sizeof(float)));  /* constant logits: uniform priors */
    HIP(hipMemset(value, 0, 2 * sizeof(float)));
    int32_t go[2] = {1, 1};
    CHECK(bo_search_begin(e, go, NULL, nn_in, NULL));
    CHECK(bo_step(e, NULL, NULL, BO_POLICY_NONE, nn_in, NULL));
    int steps = 0;
    for (;;) {
        int32_t running = 0, requested = 0;
        CHECK(bo_search_poll(e, &running, &requested, NULL, NULL));
        if (!running) break;
        CHECK(bo_step(e, policy, value, BO_POLICY_LOGITS, nn_in, NULL));
        if (++steps > 100) return 4;
    }
    static int32_t res_n[2], res_idx[2 * BO_RES_CAP], best_idx[2], best_mv[2], total[2];
    static float res_val[2 * BO_RES_CAP];
    CHECK(bo_search_result(e, res_n, res_idx, res_val, best_idx, best_mv, total, NULL));
    float first_plane_sum = 0.0f;
    static float row[BO_ROW_FLOATS];
    HIP(hipMemcpy(row, nn_in, sizeof(row), hipMemcpyDeviceToHost));
    for (int i = 98 * 64; i < 110 * 64; i++) first_plane_sum += row[i];
    printf("steps %d total %d %d pi_entries %d %d best %d %d pieces_in_block7 %.0f\n", steps, total[0], total[1], res_n[0], res_n[1],
           best_idx[0], best_idx[1], first_plane_sum);
    if (total[0] != 200 || total[1] != 200 || res_n[0] < 1 || res_n[0] > 2 || best_idx[0] < 0 || first_plane_sum != 32.0f) return 5;
    int32_t st[2];
    CHECK(bo_engine_status(e, st, NULL, NULL, NULL, NULL, NULL, NULL));
    if (st[0] || st[1]) return 6;
    bo_engine_destroy(e);
    printf("C ABI smoke ok\n");
    return 0;
}
