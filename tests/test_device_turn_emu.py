"""CPU tests (wave emulator): the ply's turn made on the DEVICE (bo_selfplay_autoturn: result -> temperature sample -> play -> begin,
csrc/bo_tree.h bo_k_turn_sample / bo_k_turn_play) plays the very games the host-made turn (bo_selfplay_turn, pinned by the oracle and the
golden games elsewhere) plays: same moves, same pi bits, same RNG streams afterwards -- across the temperature threshold
(self_play.py:66), with slots refilled in the middle of a run, with the move limit, in cohorts, and when a search needs one more evaluation
than was enqueued (the turn then does nothing and is made again)."""
import numpy as np
import pytest

import engine_harness as H
from fake_model import FakeNet


def _play(device_turn, *, G=6, plies=22, sims=40, batch=16, temperature=(3, 1.0, 0.1), max_game_moves=9, cohorts=1, n_games=11, fens=None,
          expected_evals=None):
    from betaone_amd.rollout import CohortRollout, Rollout

    with H.emulator_backend():
        kw = dict(num_simulations=sims, mcts_batch_size=batch, device="cpu", use_graph=False, rng_mode="native", policy_kind="logits",
                  temperature=temperature, max_game_moves=max_game_moves)
        ro = CohortRollout(FakeNet(), G, cohorts=cohorts, **kw) if cohorts > 1 else Rollout(FakeNet(), G, **kw)
        parts = ro.parts if cohorts > 1 else [ro]
        for p in parts:
            p.device_turn = device_turn
            if expected_evals is not None:
                p.expected_evals = expected_evals  # fewer evaluations enqueued than a search needs: every turn comes up early at first
        ro.start_games(list(range(G)), list(range(G)), [500 + g for g in range(G)], fens=[fens[g % len(fens)] for g in range(G)] if fens else None)
        nxt, fins, n_auto = [G], {}, 0

        def refill(slot):
            if nxt[0] >= n_games:
                return None
            gid = nxt[0]
            nxt[0] += 1
            return gid, 500 + gid, (fens[gid % len(fens)] if fens else None)

        for _ in range(plies):
            ro.play_ply(on_finished=lambda f: fins.__setitem__(f.game_id, f), refill=refill)
        if cohorts > 1:
            ro.drain()
        states = []
        for p in parts:
            for g in range(p.G):
                states.append(p.eng.rng_get_state(g)[1][:8].tolist() + [int(p.eng.rng_get_state(g)[2])])
        for p in parts:
            p.eng.check_status()
        ro.close()
    return {gid: (list(f.moves), [(np.asarray(i).tolist(), np.asarray(v, np.float32).view(np.uint32).tolist()) for i, v in f.pis], f.terminal, f.outcome)
            for gid, f in fins.items()}, states


def test_device_turn_plays_the_games_of_the_host_turn():
    a, sa = _play(False)
    b, sb = _play(True)
    assert len(a) >= 8 and a == b and sa == sb
    assert any(len(m) >= 6 for m, *_ in a.values())  # games crossed the temperature threshold (fullmove 3) and ran into the move limit


def test_device_turn_is_what_ran():
    """(guards the test above against silently comparing the host path with itself)"""
    from betaone_amd.rollout import Rollout

    with H.emulator_backend():
        ro = Rollout(FakeNet(), 2, num_simulations=20, mcts_batch_size=8, device="cpu", use_graph=False, rng_mode="native")
        ro.start_games([0, 1], [0, 1], [1, 2])
        assert ro._device_turn_ok()
        assert ro.ply_begin() and ro._auto and ro.eng.autoturn_ready()
        assert ro.ply_end() == 2 and not ro._auto
        ro.temperature = (30, 0.5, 0.1)  # a setting the device sampler does not cover: the host turn takes over
        assert not ro._device_turn_ok() and ro.ply_begin() and not ro._auto and ro.ply_end() == 2
        ro.close()


def test_device_turn_from_late_positions_and_in_cohorts():
    fens = ["r1bq1rk1/pp2bppp/2n1pn2/3p4/3P1B2/2PBPN2/PP1N1PPP/R2QK2R w KQ - 4 29",  # crosses fullmove 30 inside the run
            "8/8/8/4k3/8/8/4K2R/8 w - - 96 70",                                        # a fifty-move claim is near: games end by rule
            "7k/5Q2/6K1/8/8/8/8/8 b - - 0 50"]                                          # stalemate at the root: the game is over before it starts
    kw = dict(G=8, plies=10, sims=48, batch=16, temperature=(30, 1.0, 0.1), max_game_moves=300, n_games=14, fens=fens)
    a, sa = _play(False, **kw)
    b, sb = _play(True, **kw)
    assert a == b and sa == sb and len(a) >= 4
    c, sc = _play(True, cohorts=2, **kw)
    assert c == a


def test_device_turn_made_again_when_a_search_was_still_running():
    kw = dict(G=4, plies=6, sims=64, batch=16, n_games=4, max_game_moves=50)
    a, sa = _play(False, **kw)
    b, sb = _play(True, expected_evals=2, **kw)  # 1 + ceil(64 / 16) = 5 are needed: the first turns find searches running
    assert a == b and sa == sb


def test_device_turn_calls_out_of_order_and_uncovered_settings_are_refused():
    """bo_selfplay_autoturn's guards: one turn outstanding per engine (a second one, or the host-made turn, before the collect is
    BO_E_STATE); a collect without a turn is BO_E_STATE; settings the device sampler does not cover (T_initial != 1, T_final = 0, a root
    that may keep more than two children) are BO_E_CONFIG -- and Rollout routes those to the host-made turn by itself."""
    from betaone_amd import engine as E

    with H.emulator_backend():
        eng = E.Engine(2, num_simulations=16, mcts_batch_size=8, max_plies=64)
        eng.reset([0, 1])
        nn_in = H.Buf("emu", (2, 120, 8, 8))
        out = dict(n=np.zeros(2, np.int32), idx=np.zeros((2, E.RES_CAP), np.int32), val=np.zeros((2, E.RES_CAP), np.float32),
                   best_idx=np.zeros(2, np.int32), action=np.zeros(2, np.int32))
        go, mv = np.ones(2, np.int32), np.ones(2, np.int32)
        with pytest.raises(E.EngineError, match="no bo_selfplay_autoturn outstanding"):
            eng.autoturn_collect(out)
        for bad in ((30, 0.5, 0.1), (30, 1.0, 0.0)):
            assert not eng.autoturn_supported(bad)
            with pytest.raises(E.EngineError, match="TEMPERATURE"):
                eng.selfplay_autoturn(go, mv, bad, go, nn_in.ptr)
        wide = E.Engine(1, num_simulations=16, mcts_batch_size=8, widen_coeff=2.0, max_plies=64)
        assert not wide.autoturn_supported((30, 1.0, 0.1))
        with pytest.raises(E.EngineError, match="WIDEN_COEFF"):
            wide.selfplay_autoturn(go[:1], mv[:1], (30, 1.0, 0.1), go[:1], nn_in.ptr)
        wide.close()
        # a search that is still running when the turn comes up: nothing is played, the collect says so, the turn can be made again
        nl, term, goo = eng.selfplay_begin(go, nn_in.ptr)
        eng.selfplay_autoturn(goo, mv, (30, 1.0, 0.1), go, nn_in.ptr)
        with pytest.raises(E.EngineError, match="has not been collected"):
            eng.selfplay_autoturn(goo, mv, (30, 1.0, 0.1), go, nn_in.ptr)
        with pytest.raises(E.EngineError, match="autoturn_collect"):
            eng.selfplay_turn(goo, mv, (30, 1.0, 0.1), out, go, nn_in.ptr, poll_first=True, defer_noise=True)
        assert eng.autoturn_ready()
        assert eng.autoturn_collect(out) == (None, None)        # (the searches had not finished: no move was made)
        assert eng.root_info()[2].tolist() == [0, 0]            # ... both games are still at ply 0
        eng.close()
