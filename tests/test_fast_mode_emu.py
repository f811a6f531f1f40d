"""CPU tests of the FAST search mode kernels (csrc/bo_fastw.h: child-block arenas, virtual loss, tree reuse) under the wave
emulator, against the NumPy restatement tests/fast_reference.py and against structural invariants."""
import numpy as np
import pytest

import engine_cases as EC
from betaone_amd import engine as E
from engine_harness import Buf, emu_call
from fake_model import fake_logits_values
from fast_reference import FastSearcher, canonical_from_engine
from oracle import oracle as O


def softmax_eval(salt, scale=6.0):
    def fn(planes):
        logits, v = fake_logits_values(planes, scale, salt)
        x = logits.astype(np.float64)
        e = np.exp(x - x.max(axis=1, keepdims=True))
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32), v
    return fn


def run_engine_search(backend, fen, moves, sims, L, eval_fn, seed, alpha=0.1, G=1):
    kw = dict(num_simulations=sims, dirichlet_alpha=alpha, fast=True, leaves_per_step=L, max_plies=256)
    eng = emu_call(E.Engine, G, **kw) if backend == "emu" else E.Engine(G, **kw)
    eng.reset(list(range(G)), [fen] * G, [" ".join(moves) or None] * G)
    nl, term, _ = eng.root_info()
    rngs = [np.random.RandomState(seed + g) for g in range(G)]
    noise = np.zeros((G, E.MAX_LEGAL))
    for g in range(G):
        if term[g] == 0 and alpha > 0:
            noise[g, :nl[g]] = rngs[g].dirichlet([alpha] * int(nl[g]))
    nn_in, pol, val = Buf(backend, (G * L, 120, 8, 8)), Buf(backend, (G * L, E.NUM_ACTIONS)), Buf(backend, (G * L,))
    eng.search_begin([1] * G, noise if alpha > 0 else None, nn_in.ptr)
    kind, steps = E.POLICY_NONE, 0
    while True:
        eng.step(pol.ptr, val.ptr, kind, nn_in.ptr)
        steps += 1
        running, _, _ = eng.poll()
        if not running:
            break
        p, v = eval_fn(nn_in.numpy())          # evaluate every row (stale rows are ignored by the engine)
        pol.set(p); val.set(v)
        kind = E.POLICY_PROBS
        assert steps < 10000
    eng.check_status()
    return eng, noise, steps


CASES = [
    (O.STARTING_FEN, [], 64, 8),
    (O.STARTING_FEN, "e2e4 e7e5 g1f3 b8c6 f1b5".split(), 150, 16),
    ("k7/8/1K6/8/8/8/8/7R w - - 0 1", [], 120, 8),                         # mates in the tree
    (O.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6".split(), 100, 4),     # claimable repetitions in the tree
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 97 80", [], 60, 8),                      # 50-move claims in the tree
    ("R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1", [], 48, 8),   # 218 legal moves: a root run of 7 child blocks
]


def reference_for(fen, moves, fn, sims, L, **kw):
    b = O.Board(fen)
    trk = O.PyTracker(); trk.add_board(b)
    for u in moves:
        b.push(u); trk.add_board(b)
    return FastSearcher(b, trk, fn, sims, L, **kw)


def drive_search(backend, eng, G, L, eval_fn, noise, bufs=None):
    """One search in every slot of `eng` with an external evaluator (all rows are evaluated; stale rows are ignored)."""
    nn_in, pol, val = bufs or (Buf(backend, (G * L, 120, 8, 8)), Buf(backend, (G * L, E.NUM_ACTIONS)), Buf(backend, (G * L,)))
    eng.search_begin([1] * G, noise, nn_in.ptr)
    kind, steps = E.POLICY_NONE, 0
    while True:
        eng.step(pol.ptr, val.ptr, kind, nn_in.ptr)
        steps += 1
        running, _, _ = eng.poll()
        if not running:
            break
        p, v = eval_fn(nn_in.numpy())
        pol.set(p); val.set(v)
        kind = E.POLICY_PROBS
        assert steps < 10000
    eng.check_status()
    return steps, (nn_in, pol, val)


def test_a_full_arena_narrows_the_search_and_is_not_a_fault():
    """Fast mode, an arena far too small for the search (bo_fastw.h: a leaf whose run does not fit stays unexpanded, its value is still
    backed up): the search finishes all its simulations, pi is a distribution over legal moves, the slot carries the soft
    'node overflow' bit -- check_status counts it and does not raise (a 16-ply run of 32 768 games hits this in a handful of slots)."""
    sims, L = 150, 8
    kw = dict(num_simulations=sims, dirichlet_alpha=0.0, fast=True, leaves_per_step=L, max_plies=256, fast_arena_granules=64)
    eng = emu_call(E.Engine, 1, **kw)
    eng.reset([0], [O.STARTING_FEN], [None])
    steps, _ = drive_search("emu", eng, 1, L, softmax_eval(5), None)
    assert eng.check_status() == 1 and eng.status_bits()[0] == E.ST_NODE_OVERFLOW
    res = eng.result()
    n, idx, val = int(res["n"][0]), res["idx"][0], res["val"][0]
    assert 1 <= n <= 20 and abs(float(val[:n].sum()) - 1.0) < 1e-5 and steps >= sims // L
    legal = {O.move_to_index(m) for m in O.Board(O.STARTING_FEN).legal_moves()}
    assert set(int(i) for i in idx[:n]) <= legal
    eng.close()
    # the reference-semantics engine keeps treating the bit as a fault
    assert E.Engine.soft_status_bits(type("X", (), {"fast": False})()) == 0


@pytest.mark.parametrize("fen,moves,sims,L", CASES)
def test_fast_kernels_match_numpy_restatement(fen, moves, sims, L):
    fn = softmax_eval(7)
    eng, noise, _ = run_engine_search("emu", fen, moves, sims, L, fn, seed=3)
    ref = reference_for(fen, moves, fn, sims, L)
    ref.search(noise[0])
    assert canonical_from_engine(eng.debug_tree(0), E.move_to_uci) == ref.canonical()
    res = eng.result()
    visits = [v for _m, v in ref.visits()]
    n = int(res["n"][0])
    assert n == sum(v > 0 for v in visits) and int(res["total"][0]) == sum(visits) == sims
    best = int(np.argmax(visits))
    assert E.move_to_uci(int(res["best_move"][0])) == ref.visits()[best][0]
    assert abs(float(res["val"][0, :n].sum()) - 1.0) < 1e-6
    st = eng.status()
    assert int(st["evals"][0]) == ref.n_evals and int(st["term_sims"][0]) == ref.n_term_sims


@pytest.mark.parametrize("reuse", [True, False])
def test_fast_tree_reuse_between_moves_matches_restatement(reuse):
    """Four consecutive searches of one game; after each the most visited move is played and the played child's subtree
    is the next search's tree (bo_k_fw_reroot: breadth-first compaction into the game's other arena) -- or, with reuse
    switched off, a fresh root."""
    fen, moves, sims, L = O.STARTING_FEN, "d2d4 d7d5 c2c4".split(), 96, 8
    fn = softmax_eval(5)
    kw = dict(num_simulations=sims, dirichlet_alpha=0.1, fast=True, leaves_per_step=L, max_plies=256)
    eng = emu_call(E.Engine, 1, **kw)
    eng.fast_options(tree_reuse=reuse)
    eng.reset([0], [fen], [" ".join(moves)])
    ref = reference_for(fen, moves, fn, sims, L, reuse=reuse)
    rng, bufs, kept = np.random.RandomState(11), None, []
    for ply in range(4):
        nl, term, _ = eng.root_info()
        assert term[0] == 0
        noise = np.zeros((1, E.MAX_LEGAL))
        noise[0, :nl[0]] = rng.dirichlet([0.1] * int(nl[0]))
        kept.append(eng.debug_tree(0)[0]["n"])                      # visits the new root starts with
        _, bufs = drive_search("emu", eng, 1, L, fn, noise, bufs)
        ref.search(noise[0])
        got = canonical_from_engine(eng.debug_tree(0), E.move_to_uci)
        assert got == ref.canonical(), ply
        res = eng.result()
        assert int(res["total"][0]) == sum(v for _m, v in ref.visits()) >= sims
        best = E.move_to_uci(int(res["best_move"][0]))
        eng.play(np.array([-2], dtype=np.int32))                    # play res_best_mv; the engine re-roots
        ref.play(best)
        after = canonical_from_engine(eng.debug_tree(0), E.move_to_uci)
        assert after == ref.canonical(), ("after re-rooting", ply)   # the kept subtree, bit for bit (w, priors, states)
    assert (max(kept[1:]) > 1) == reuse                              # reuse really carried visits over
    fs = eng.fast_stats()
    assert fs["granules_read"][0] > 0 and 1 <= fs["arena_granules"][0] < eng.cfg.num_simulations * 32


def test_fast_mode_invariants_many_games():
    G, sims, L = 6, 200, 16
    eng, _, steps = run_engine_search("emu", O.STARTING_FEN, ["d2d4"], sims, L, softmax_eval(11), seed=1, G=G)
    assert steps <= 2 + (sims + L - 1) // L + 2
    for g in range(G):
        t = eng.debug_tree(g)
        root = t[0]
        kids = t[root["first_child"]:root["first_child"] + root["n_children"]]
        assert root["n_children"] == 20 and sum(k["n"] for k in kids) == sims and root["n"] == sims + 1
        assert abs(sum(float(k["prior"]) for k in kids) - 1.0) < 1e-5
        for nd in t[1:]:
            assert nd["n"] >= 0 and abs(float(nd["q"])) <= nd["n"] + 1e-4        # no virtual loss left behind
            if nd["n_children"]:
                ch = t[nd["first_child"]:nd["first_child"] + nd["n_children"]]
                assert nd["n"] == 1 + sum(c["n"] for c in ch)                      # expanded by 1 visit, rest went below
    # different seeds (noise) -> different games, same seed -> identical
    a = [tuple((k["n"]) for k in eng.debug_tree(g)[1:21]) for g in range(G)]
    assert len(set(a)) > 1


def test_fast_rollout_refilled_slot_records_the_pis_of_its_own_game():
    """A slot refilled inside a ply sits that ply out: the new game's first pi belongs to the NEXT step, not to the row its
    slot's previous occupant left behind (round-2 advisor finding: every (state, pi) record of such a game was shifted by
    one ply).  Staggered starts + refill in a 2-slot rollout; every game's (moves, pis) must equal a solo run of its id."""
    import torch
    from betaone_amd.rollout import Rollout
    from fake_model import FakeNet

    class Net(torch.nn.Module):
        def forward(self, x):
            return FakeNet(scale=2.0, salt=3)(x)

    start = ["k7/8/1K6/8/8/8/8/7R w - - 96 60", None, "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 30"]

    def run(slots, ids, stagger):
        ro = emu_call(Rollout, Net(), slots, num_simulations=12, mcts_batch_size=8, max_game_moves=4, device="cpu", use_graph=False,
                      rng_mode="native", fast=True, leaves_per_step=4)
        todo, fins = list(ids), {}

        def refill(_slot):
            if not todo:
                return None
            gid = todo.pop(0)
            return gid, 50 + gid, start[gid % 3]

        first = [todo.pop(0) for _ in range(min(slots, len(todo)))]
        ro.start_games([0], [first[0]], [50 + first[0]], [start[first[0] % 3]])
        for _ in range(stagger):  # slot 1 enters two plies later: the two games never end in the same ply
            ro.play_ply(on_finished=lambda f: fins.__setitem__(f.game_id, f), refill=refill)
        if len(first) > 1:
            ro.start_games([1], [first[1]], [50 + first[1]], [start[first[1] % 3]])
        while any(g is not None for g in ro.games):
            ro.play_ply(on_finished=lambda f: fins.__setitem__(f.game_id, f), refill=refill)
        ro.close()
        return fins

    many = run(2, list(range(6)), 2)
    assert sorted(many) == list(range(6))
    for gid in range(6):
        solo = run(1, [gid], 0)[gid]
        got = many[gid]
        assert got.moves == solo.moves and got.terminal == solo.terminal, gid
        assert len(got.pis) == len(solo.pis) == len(got.moves), gid
        for (i1, v1), (i2, v2) in zip(got.pis, solo.pis):
            assert i1.tolist() == i2.tolist() and v1.tolist() == v2.tolist(), gid


MULTI = [
    (O.STARTING_FEN, []),
    (O.STARTING_FEN, "e2e4 e7e5 g1f3 b8c6 f1b5".split()),
    ("k7/8/1K6/8/8/8/8/7R w - - 0 1", []),                               # mates in the tree: known-terminal simulations
    (O.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6".split()),           # claimable repetitions
    ("R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1", []),         # 218 legal moves: a root run of 28 granules
    ("8/8/4k3/8/8/3K4/8/6R1 w - - 97 80", []),                            # 50-move claims
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", []),  # 48 legal moves: two records per lane
]


def run_multi(backend, L, sims, opts, seed=5):
    """Every game slot holds a different position (the half-waves and the games interleaved per half-wave of bo_k_fw_select
    then diverge in depth, run length and end of descent); returns engine + per-game noise."""
    G = len(MULTI)
    kw = dict(num_simulations=sims, dirichlet_alpha=0.1, fast=True, leaves_per_step=L, max_plies=256)
    eng = emu_call(E.Engine, G, **kw) if backend == "emu" else E.Engine(G, **kw)
    eng.fast_options(**opts)
    eng.reset(list(range(G)), [m[0] for m in MULTI], [" ".join(m[1]) or None for m in MULTI])
    nl, term, _ = eng.root_info()
    noise = np.zeros((G, E.MAX_LEGAL))
    for g in range(G):
        noise[g, :nl[g]] = np.random.RandomState(seed + g).dirichlet([0.1] * int(nl[g]))
    fn = softmax_eval(13)
    drive_search(backend, eng, G, L, fn, noise)
    return eng, noise, fn


def check_multi(backend, L, sims, opts):
    eng, noise, fn = run_multi(backend, L, sims, opts)
    st = eng.status()
    for g, (fen, moves) in enumerate(MULTI):
        ref = reference_for(fen, moves, fn, sims, L)
        ref.search(noise[g])
        assert canonical_from_engine(eng.debug_tree(g), E.move_to_uci) == ref.canonical(), (g, L, opts)
        assert int(st["evals"][g]) == ref.n_evals and int(st["term_sims"][g]) == ref.n_term_sims, (g, L, opts)
    fs = eng.fast_stats()
    assert (fs["granules_read"] > 0).all() and (fs["path_nodes"] >= sims).all()
    return eng


@pytest.mark.parametrize("L,sims,opts", [
    (4, 60, dict(select_flags=8)),       # one lane per game
    (4, 80, dict(select_flags=9)),
    (4, 130, dict(select_flags=8)),
    (4, 90, dict(select_flags=16)),      # eight lanes per game
    (4, 90, dict(select_flags=18)),
    (3, 60, dict(select_flags=17)),
    (7, 84, dict(select_flags=16)),
    (8, 96, dict(select_flags=18)),
    (4, 90, dict(select_flags=32)),      # four lanes per game: ten records per lane, two path depths per lane in the backup
    (4, 130, dict(select_flags=34)),
    (3, 60, dict(select_flags=32)),
    (7, 84, dict(select_flags=32)),
    (8, 96, dict(select_flags=32)),
    (4, 70, dict()),                      # the defaults
    (4, 60, dict(games_per_halfwave=4, select_flags=2)),
    (4, 60, dict(games_per_halfwave=4, select_flags=0)),
    (3, 50, dict(games_per_halfwave=2, select_flags=3)),
    (2, 40, dict(games_per_halfwave=2, select_flags=0)),
    (7, 70, dict(select_flags=2)),
    (8, 64, dict(select_flags=0)),
    (13, 90, dict(select_flags=2)),
    (33, 99, dict(select_flags=2)),
    (64, 128, dict(select_flags=0)),
])
def test_fast_select_variants_match_restatement_with_a_different_position_in_every_slot(L, sims, opts):
    check_multi("emu", L, sims, opts)
