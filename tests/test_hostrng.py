"""CPU tests: the engine's native per-game random streams (betaone_amd/csrc/bo_hostrng.h) against
numpy.random.RandomState draw for draw -- seeding, random_sample, legacy dirichlet, temperature sampling."""
import numpy as np
import pytest

from betaone_amd import engine as E, sampling
from engine_harness import emu_call


@pytest.fixture(scope="module")
def eng():
    return emu_call(E.Engine, 4, num_simulations=10)


@pytest.mark.parametrize("seed", [0, 1, 7, 12345, 2**31 + 5, 2**32 - 1])
def test_seeding_and_state_roundtrip_match_randomstate(eng, seed):
    eng.rng_seed(1, seed)
    st = eng.rng_get_state(1)
    ref = np.random.RandomState(seed).get_state()
    assert np.array_equal(st[1], ref[1]) and st[2:] == tuple(ref[2:])
    rs = np.random.RandomState(0)
    rs.set_state(st)
    assert rs.random_sample() == np.random.RandomState(seed).random_sample()


def _native_stream(eng, slot):
    rs = np.random.RandomState(0)
    rs.set_state(eng.rng_get_state(slot))
    return rs


def test_sampling_matches_numpy_draw_for_draw(eng):
    gen = np.random.RandomState(99)
    out = dict(n=np.zeros(4, np.int32), idx=np.zeros((4, E.RES_CAP), np.int32), val=np.zeros((4, E.RES_CAP), np.float32),
               best_idx=np.zeros(4, np.int32), action=np.zeros(4, np.int32))
    import ctypes as C
    L = eng.lib
    for trial in range(3000):
        seed = int(gen.randint(1 << 31))
        n = int(gen.randint(1, 3))
        idx = gen.choice(4672, n, replace=False).astype(np.int32)
        tot = int(gen.randint(1, 801))
        if n == 2 and tot > 1:
            a = int(gen.randint(1, tot))
            cnt = [a, tot - a]
        else:
            cnt, idx = [tot], idx[:1]
        val = np.array([np.float32(c / tot) for c in cnt], dtype=np.float32)
        mv = int(gen.randint(1, 70))
        ref = np.random.RandomState(seed)
        exp = sampling.select_move_with_temperature(sampling.dense_pi(idx, val), mv, ref)
        # drive hr_select_action through the public entry point is not possible without a finished search:
        # use the stream check instead -- seed, sample via the sparse Python path on the exported state
        eng.rng_seed(0, seed)
        rs = _native_stream(eng, 0)
        got = sampling.select_action_sparse(idx, val, mv, rs)
        assert got == exp and rs.random_sample() == ref.random_sample()


@pytest.mark.parametrize("alpha", [0.1, 0.3, 1.0, 2.5])
def test_dirichlet_matches_numpy(eng, alpha):
    """bo_selfplay_begin draws the root noise natively: compare the stream position afterwards and the noise via
    a search's visible effect is covered elsewhere; here: state equality after the same number of draws."""
    e = emu_call(E.Engine, 3, num_simulations=10, dirichlet_alpha=alpha)
    e.reset([0, 1, 2], [None, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", None])
    nn_in = np.zeros((3, 120, 8, 8), dtype=np.float32)
    for g, seed in enumerate([5, 6, 7]):
        e.rng_seed(g, seed)
    nl, term, go = e.selfplay_begin([1, 1, 0], nn_in.ctypes.data)
    assert list(go) == [1, 1, 0] and list(nl[:2]) == [20, 48]
    for g, seed in enumerate([5, 6, 7]):
        ref = np.random.RandomState(seed)
        if go[g]:
            ref.dirichlet([alpha] * int(nl[g]))
        st = e.rng_get_state(g)
        rs = ref.get_state()
        assert np.array_equal(st[1], rs[1]) and st[2] == rs[2] and st[3] == rs[3] and st[4] == rs[4]
    e.close()


def test_native_rollout_equals_python_rollout_and_oracle():
    """Whole games: rng_mode='native' (streams inside the engine) == rng_mode='python' (numpy RandomState per game)
    == the CPU oracle, including games that cross the temperature threshold (fullmove >= 30)."""
    import torch
    from betaone_amd.rollout import Rollout
    from fake_model import FakeNet, fake_logits_values
    from oracle import oracle as O

    class Net(torch.nn.Module):
        def forward(self, x):
            return FakeNet(scale=0.0, salt=31)(x)

    fens = [None, None, "r1bqkbnr/pppp1ppp/2n5/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R w KQkq - 2 29", "k7/8/1K6/8/8/8/8/7R w - - 90 60"]
    seeds = [3, 4, 5, 6]

    def play(mode):
        ro = emu_call(Rollout, Net(), 4, num_simulations=40, mcts_batch_size=16, max_game_moves=7, device="cpu", use_graph=False,
                      rng_mode=mode)
        rngs = seeds if mode == "native" else [np.random.RandomState(s) for s in seeds]
        ro.start_games([0, 1, 2, 3], [0, 1, 2, 3], rngs, fens)
        fins = {}
        calls = [0, 0]  # play_ply calls, while_searching calls; finished games must already be reported when the hook runs
        seen_at_hook = []
        while any(g is not None for g in ro.games):
            calls[0] += 1
            ro.play_ply(on_finished=lambda f: fins.__setitem__(f.game_id, f),
                        while_searching=lambda: (calls.__setitem__(1, calls[1] + 1), seen_at_hook.append(len(fins))))
            assert seen_at_hook[-1] == len(fins)  # nothing is reported after the hook of the same ply
        assert calls[0] == calls[1]  # exactly once per ply, in every exit path of play_ply
        ro.close()
        return fins

    a, b = play("native"), play("python")
    assert sorted(a) == sorted(b) == [0, 1, 2, 3]
    for gid in a:
        assert a[gid].moves == b[gid].moves and a[gid].terminal == b[gid].terminal
        assert len(a[gid].pis) == len(b[gid].pis)
        for (i1, v1), (i2, v2) in zip(a[gid].pis, b[gid].pis):
            assert i1.tolist() == i2.tolist() and v1.tolist() == v2.tolist()

    def uniform_eval(planes):
        _, v = fake_logits_values(planes, 0.0, 31)
        return np.full((planes.shape[0], 4672), np.float32(1.0) / np.float32(4672.0), dtype=np.float32), v

    for gid in (0, 2, 3):
        ref = O.self_play(uniform_eval, np.random.RandomState(seeds[gid]),
                          O.default_config(num_simulations=40, batch_size=16, max_game_moves=7), start_fen=fens[gid] or "")
        assert [E.move_to_uci(m) for m in a[gid].moves] == [O.move_to_uci(m) for m in ref["moves"]]


def test_native_rollout_with_slot_recycling_equals_python_rollout():
    """Finished games hand their slot to the next game id (refill) while the other slots' searches were already begun by
    the previous bo_selfplay_turn: the games played must not depend on rng_mode."""
    import torch
    from betaone_amd.rollout import Rollout
    from fake_model import FakeNet

    class Net(torch.nn.Module):
        def forward(self, x):
            return FakeNet(scale=0.5, salt=7)(x)

    start = ["k7/8/1K6/8/8/8/8/7R w - - 96 60", None, "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 30"]

    def play(mode):
        ro = emu_call(Rollout, Net(), 3, num_simulations=24, mcts_batch_size=8, max_game_moves=5, device="cpu", use_graph=False,
                      rng_mode=mode)
        mk = (lambda s: s) if mode == "native" else (lambda s: np.random.RandomState(s))
        ro.start_games([0, 1, 2], [0, 1, 2], [mk(10), mk(11), mk(12)], start)
        nxt, fins = [3], {}

        def refill(_slot):
            if nxt[0] >= 8:
                return None
            gid = nxt[0]
            nxt[0] += 1
            return gid, mk(10 + gid), start[gid % 3]

        while any(g is not None for g in ro.games):
            ro.play_ply(on_finished=lambda f: fins.__setitem__(f.game_id, f), refill=refill)
        ro.close()
        return fins

    a, b = play("native"), play("python")
    assert sorted(a) == sorted(b) == list(range(8))
    for gid in a:
        assert a[gid].moves == b[gid].moves and a[gid].terminal == b[gid].terminal, gid
        for (i1, v1), (i2, v2) in zip(a[gid].pis, b[gid].pis):
            assert i1.tolist() == i2.tolist() and v1.tolist() == v2.tolist()


def test_turn_without_waiting_reports_the_roots_later_and_refuses_calls_out_of_order():
    """bo_selfplay_turn(flag 4): the next searches are begun on the device without a host round trip; bo_selfplay_begun then
    returns what bo_selfplay_begin would have (n_legal, terminal, go) -- compared here with the waiting form of the same turn on
    a second engine -- and calls out of order are refused (BO_E_STATE), not silently wrong."""
    import torch
    from fake_model import fake_logits_values

    def search_all(en, nn_in):
        for _ in range(12):
            running, _, _ = en.poll(0, want_mask=False)
            if running == 0:
                return
            planes = nn_in.numpy().copy()
            logits, values = fake_logits_values(planes, 0.0, 5)
            probs = torch.softmax(torch.from_numpy(logits), dim=1).numpy().astype(np.float32)
            en.step(probs.ctypes.data, values.astype(np.float32).ctypes.data, E.POLICY_PROBS, nn_in.data_ptr(), 0)
        raise AssertionError("search did not finish")

    outs = []
    for lazy in (False, True):
        en = emu_call(E.Engine, 3, num_simulations=20, mcts_batch_size=8)
        nn_in = torch.zeros((3, E.INPUT_CHANNELS, 8, 8), dtype=torch.float32)
        en.reset([0, 1, 2], ["k7/8/1K6/8/8/8/8/7R w - - 0 1", None, "7k/5Q2/6K1/8/8/8/8/8 w - - 0 1"], None)
        for g in range(3):
            en.rng_seed(g, 100 + g)
        want = np.ones(3, dtype=np.int32)
        nl, term, go = en.selfplay_begin(want, nn_in.data_ptr(), 0)
        assert go.tolist() == [1, 1, 1]
        search_all(en, nn_in)
        out = dict(n=np.zeros(3, np.int32), idx=np.zeros((3, E.RES_CAP), np.int32), val=np.zeros((3, E.RES_CAP), np.float32),
                   best_idx=np.zeros(3, np.int32), action=np.zeros(3, np.int32))
        o, begun = en.selfplay_turn(go, np.ones(3, np.int32), (30, 1.0, 0.1), out, want, nn_in.data_ptr(), 0, defer_noise=True,
                                    poll_first=True, lazy_begin=lazy)
        assert o is not None
        if lazy:
            assert begun is E.LAZY_BEGIN
            with pytest.raises(E.EngineError):   # the roots of the turn have not been collected yet
                en.selfplay_begin(want, nn_in.data_ptr(), 0)
            with pytest.raises(E.EngineError):
                en.selfplay_noise(0)
            begun = en.selfplay_begun()
            with pytest.raises(E.EngineError):   # ... and only once
                en.selfplay_begun()
        en.selfplay_noise(0)
        outs.append((o["action"].copy(), [b.copy() for b in begun], nn_in.numpy().copy()))
        en.close()
    (a0, b0, x0), (a1, b1, x1) = outs
    assert a0.tolist() == a1.tolist()
    for u, v in zip(b0, b1):
        assert u.tolist() == v.tolist()
    assert np.array_equal(x0, x1)   # the same root planes are on their way to the net
