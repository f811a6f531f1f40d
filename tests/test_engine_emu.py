"""CPU tests of the ENGINE'S DEVICE CODE (betaone_amd/csrc) under the 64-lane wave emulator: the
same sources hipcc compiles for gfx950, executed lane by lane on the CPU.  They check kernel logic
(and memory safety) before any GPU minute is spent; the `-m gpu` tests repeat them on the product
library.  Nothing here is a product path."""
import numpy as np
import pytest

import engine_cases as EC


@pytest.fixture(scope="module")
def backend():
    return "emu"


def test_movegen_matches_oracle_order(backend):
    EC.check_movegen_random_positions(backend, n_games=12, max_plies=60, seed=0)


def test_movegen_special_positions(backend):
    EC.check_movegen_special(backend)


@pytest.mark.parametrize("name", EC.SEARCH_NAMES)
def test_search_matches_reference_trace(backend, name):
    EC.check_golden_search(backend, name)


@pytest.mark.parametrize("name", EC.GAME_NAMES)
def test_self_play_game_matches_reference(backend, name):
    EC.check_golden_game(backend, name)


def test_many_games_in_lockstep_match_oracle(backend):
    EC.check_multi_game_vs_oracle(backend, n_games=5, plies=6, sims=60, batch=16)


def test_full_games_to_termination_match_oracle(backend):
    """Whole games until is_game_over(claim_draw=True): long-game logic (claimable draws in the REAL game, tracker and
    position-stack growth, end-of-game tracker in the training encodings) against the oracle."""
    EC.check_full_games_vs_oracle(backend, n_games=4, sims=12, batch=8)


def test_edge_cases_maximum_sizes_and_error_paths(backend):
    EC.check_edge_cases(backend)


@pytest.mark.parametrize("cfg", EC.SEARCH_CONFIG_SWEEP, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_search_configuration_sweep_matches_oracle(backend, cfg):
    """Batch sizes around the simulation count, unusual CPUCT / widening / Dirichlet settings: moves, pi and states bit-exact."""
    EC.check_multi_game_vs_oracle(backend, n_games=2, plies=3, **cfg)


def test_games_from_special_start_positions_match_oracle(backend):
    EC.check_games_from_positions_vs_oracle(backend)


@pytest.mark.parametrize("case", EC.LONG_TERMINAL_RUN_CASES, ids=lambda c: f"{c[0].split()[0][:12]}-{c[2]}")
def test_long_runs_of_terminal_simulations_match_oracle(backend, case):
    EC.check_long_terminal_runs_vs_oracle(backend, case)


def test_watched_status_word_arrives_with_the_result_block(backend):
    EC.check_watched_status_word(backend)


def test_step_heads_value_is_the_rows_kernels_value(backend):
    """bo_step_heads on the emulator: the value the step kernel makes from value_fc1's partial sums is the rows kernel's operation
    sequence (16 chunks added in order, bias, ReLU, times value_fc2's weight, a 32..1 butterfly per 64 hidden units, (w0 + w1) + (w2 + w3),
    bias, tanhf) -- restated here in numpy float32 with libm's tanhf -- and the search it drives is the search bo_step drives with
    that value and the same logits.  (That the GPU's softmax and tanhf give bo_k_heads_rows' bits is tests/test_engine_gpu.py's part.)"""
    import ctypes
    from betaone_amd import engine as E
    from engine_harness import Buf, canonical_tree, make_engine

    libm = ctypes.CDLL("libm.so.6")
    libm.tanhf.restype, libm.tanhf.argtypes = ctypes.c_float, [ctypes.c_float]
    G, ROWS = 3, 5
    rs = np.random.RandomState(3)
    b1, w2, b2 = (rs.randn(256) * 0.1).astype(np.float32), (rs.randn(256) * 0.3).astype(np.float32), np.float32([0.05])

    def rows_value(part):  # part [16, 256] float32
        h = np.zeros(256, np.float32)
        for ks in range(16):
            h = h + part[ks]
        h = h + b1
        h = np.where(h > 0, h, np.float32(0)) * w2
        s = []
        for w in range(4):
            v = h[64 * w:64 * w + 64].copy()
            m = 32
            while m >= 1:
                v = v + v[np.arange(64) ^ m]
                m >>= 1
            s.append(v[0])
        return np.float32(libm.tanhf(np.float32((s[0] + s[1]) + (s[2] + s[3]) + b2[0])))

    cfg = dict(num_simulations=150, batch_size=32, dirichlet_alpha=0.0)
    engs = [make_engine(backend, G, cfg) for _ in range(2)]
    fens = [None, "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 40", "8/5k2/8/8/8/2K5/8/4R3 b - - 90 75"]
    bufs = []
    for e in engs:
        e.reset(list(range(G)), fens)
        bufs.append(dict(nn=Buf(backend, (G, 120, 8, 8)), pol=Buf(backend, (G, E.NUM_ACTIONS)), val=Buf(backend, (G,)), part=Buf(backend, (16, ROWS, 256)),
                         b1=Buf(backend, (256,)), w2=Buf(backend, (256,)), b2=Buf(backend, (1,))))
        bufs[-1]["b1"].set(b1); bufs[-1]["w2"].set(w2); bufs[-1]["b2"].set(b2)
    go = np.ones(G, np.int32)
    n_eval = 0
    for k, e in enumerate(engs):
        B = bufs[k]
        e.search_begin(go, None, B["nn"].ptr)
        e.step(0, 0, E.POLICY_NONE, B["nn"].ptr)
        while e.poll()[0]:
            planes = B["nn"].numpy()
            logits = np.zeros((G, E.NUM_ACTIONS), np.float32)
            part = np.zeros((16, ROWS, 256), np.float32)
            for g in range(G):
                r = np.random.RandomState(int(np.abs(planes[g]).sum() * 1000) % (2 ** 31) + 17 * g)
                logits[g] = (r.randn(E.NUM_ACTIONS) * 2).astype(np.float32)
                part[:, g] = (r.randn(16, 256) * 0.2).astype(np.float32)
            B["pol"].set(logits)
            if k == 0:
                B["val"].set(np.array([rows_value(part[:, g]) for g in range(G)], np.float32))
                e.step(B["pol"].ptr, B["val"].ptr, E.POLICY_LOGITS, B["nn"].ptr)
            else:
                B["part"].set(part)
                e.step_heads(B["pol"].ptr, B["part"].ptr, B["b1"].ptr, B["w2"].ptr, B["b2"].ptr, ROWS, B["nn"].ptr)
            n_eval += 1
        e.check_status()
    assert n_eval >= 12
    ra, rb = engs[0].result(), engs[1].result()
    for g in range(G):
        assert canonical_tree(engs[0].debug_tree(g)) == canonical_tree(engs[1].debug_tree(g)), g
    assert ra["idx"].tolist() == rb["idx"].tolist() and ra["val"].view(np.uint32).tolist() == rb["val"].view(np.uint32).tolist()
    with pytest.raises(E.EngineError, match="rows >= G"):
        engs[1].step_heads(bufs[1]["pol"].ptr, bufs[1]["part"].ptr, bufs[1]["b1"].ptr, bufs[1]["w2"].ptr, bufs[1]["b2"].ptr, G - 1, bufs[1]["nn"].ptr)
    for e in engs:
        e.close()
