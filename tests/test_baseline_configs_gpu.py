"""BASELINE.json configs[2], [3], [4] at their full per-GPU workloads on the product path (cuda:0, libbetaone_hip.so,
hipGraph on, the hand-written evaluate-stage kernels), each checked against the CPU oracle from the recorded
(planes -> softmax probabilities, value) seam onward: moves, pi and trees bit for bit.

The evaluations run inside captured graphs, so the seam is recorded from OUTSIDE the graph: the NN input rows of the
watched slots are read before a replay, the probabilities / values the step kernel consumed after it."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def env():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    sys.path.insert(0, os.path.join(ROOT, "oracle", "shim"))
    from betaone_amd import dropin
    from betaone_amd import engine as E

    E.load_hip_library()
    dropin.install()
    import config

    import mcts

    keys = ("RESIDUAL_BLOCKS", "SE_RESIDUAL_BLOCKS", "CONV_FILTERS", "NUM_SIMULATIONS", "MCTS_BATCH_SIZE", "POLICY_SOFTMAX", "DATA_DIR",
            "MAX_GAME_MOVES")
    saved = {k: getattr(config, k) for k in keys}
    mcts.stop_event.clear()
    yield config
    mcts.stop_event.clear()
    for k, v in saved.items():
        setattr(config, k, v)


class SeamRecorder:
    """Wraps the two evaluate entry points of a native-RNG Rollout; records (planes key -> probs, value) of `watch` slots."""

    def __init__(self, ro, watch):
        from fake_model import planes_key

        self.ro, self.watch, self.seam, self.key = ro, list(watch), {}, planes_key
        self._step, self._fwd = ro._eval_and_step, ro._forward_only
        ro._eval_and_step, ro._forward_only = self.eval_and_step, self.forward_only
        # a search's iterations one graph launch at a time, so that every evaluation's planes can be read back
        # (test_iterations_in_one_graph_launch_play_the_same_games covers the n-iterations-per-launch form)
        ro._eval_and_step_n = lambda n: [self.eval_and_step() for _ in range(n)]

    def _planes(self):
        return self.ro.nn_in[self.watch].cpu().numpy()

    def _store(self, planes, probs, value):
        probs, value = probs[self.watch].float().cpu().numpy(), value.reshape(-1)[self.watch].float().cpu().numpy()
        for i in range(len(self.watch)):
            self.seam[self.key(planes[i])] = (probs[i].copy(), np.float32(value[i]))

    def eval_and_step(self):
        planes = self._planes()
        self._step()
        self._store(planes, self.ro._logits, self.ro._value)

    def forward_only(self):
        planes = self._planes()
        self._fwd()
        self._store(planes, self.ro._f_logits, self.ro._f_value)

    def eval_fn(self, planes):
        probs = np.zeros((planes.shape[0], 4672), np.float32)
        vals = np.zeros(planes.shape[0], np.float32)
        for i in range(planes.shape[0]):
            probs[i], vals[i] = self.seam[self.key(planes[i])]
        return probs, vals


def _check_games_against_oracle(ro, rec, watch, sims, plies):
    from betaone_amd import engine as E
    from oracle import oracle as O

    G = ro.G
    ro.eng.check_status()
    st = ro.eng.status()
    assert (st["evals"] >= plies * 2).all() and (st["flushes"] >= plies).all()
    for step in range(ro._step):  # size-independent invariants for EVERY game: pi is a distribution over <= 2 moves (E2)
        went, n, idx, val = ro._hist[step]
        assert went.all()
        assert (n >= 1).all() and (n <= 2).all()
        sums = np.array([val[g, :n[g]].sum() for g in range(G)])
        assert np.abs(sums - 1.0).max() < 1e-6
    games = {g: ro._finish(g, 0) for g in watch}
    for g in watch:
        ref = O.self_play(rec.eval_fn, np.random.RandomState(g), O.default_config(num_simulations=sims, max_game_moves=plies))
        assert [O.move_to_uci(m) for m in ref["moves"]] == [E.move_to_uci(m) for m in games[g].moves], g
        assert len(games[g].pis) == plies
        for (_, rpi, _), (idx, val) in zip(ref["records"], games[g].pis):
            assert sorted(np.nonzero(rpi)[0].tolist()) == sorted(idx.tolist())
            for i, v in zip(idx, val):
                assert np.float32(rpi[i]).view(np.uint32) == np.float32(v).view(np.uint32)


LATE_GAME_FENS = [
    "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 40",          # back-rank mate in one: terminal bursts from the first search on
    "7k/5Q2/5K2/8/8/8/8/8 w - - 10 70",                 # several mates in one side by side, stalemating moves among them
    "8/8/4k3/8/8/3K4/8/6R1 w - - 98 80",                # halfmove clock 98: claimable fifty-move draws in the tree
    "k7/8/1K6/8/8/8/8/7R w - - 96 60",                  # mate in one AND the 50-move claim close
    "8/5k2/8/8/8/2K5/8/4R3 b - - 90 75",                # black to move, long reversible chains (repetition claims)
    "r1bq1rk1/pp2bppp/2n1pn2/3p4/3P1B2/2PBPN2/PP1N1PPP/R2QK2R w KQ - 4 29",  # a middlegame that crosses the temperature threshold (fullmove 30) after three plies
]


_LAST_RUN = {}


def _bench_like_run(net, G, sims, max_game_moves, preroll, steps, watch, record, cohorts=1, cu_masks=None, policy_kind="probs"):
    """bench.py's own step loop (bench.Driver: staggered pre-roll, finished games exported and their slots refilled on the side
    stream inside the run, n-iteration graphs, native RNG), with every 8th refilled game starting from a late-game position.
    record=True: the same run with the evaluate stage launched one iteration at a time and the seam of the `watch` slots read
    back.  Returns (finished games by id, start FEN by id, recorder)."""
    import bench
    from betaone_amd.rollout import CohortRollout, Rollout

    kw = dict(num_simulations=sims, mcts_batch_size=96, device="cuda:0", use_graph=True, rng_mode="native", policy_kind=policy_kind,
              max_game_moves=max_game_moves)
    ro = CohortRollout(net, G, cohorts=cohorts, cu_masks=cu_masks, **kw) if cohorts > 1 else Rollout(net, G, **kw)
    _LAST_RUN["step_tail"] = all(p.step_tail for p in ro.parts) if cohorts > 1 else ro.step_tail
    rec = SeamRecorder(ro, watch) if record else None
    fens = {}

    class Drv(bench.Driver):
        def refill(self, _slot):
            i = self.new_id()
            fens[i] = LATE_GAME_FENS[(i // 8) % len(LATE_GAME_FENS)] if i % 8 == 5 else None
            return i, i, fens[i]

    drv = Drv(ro, 0, 1, None)
    fins = {}
    drv.on_finished = lambda f: fins.__setitem__(f.game_id, f)
    drv.preroll(preroll, G)
    for _ in range(steps):
        drv.step()
    if cohorts > 1:
        ro.drain()
    ro.eng.check_status()
    slot_of = {gid: f.slot for gid, f in fins.items()}
    n_graphs = sum(len(p._graphs_n) for p in ro.parts) if cohorts > 1 else len(ro._graphs_n)
    ro.close()
    return fins, fens, rec, slot_of, n_graphs


def test_two_path_terminal_burst_plays_the_same_games_as_the_one_path_form(env, monkeypatch):
    """terminal_burst (csrc/bo_tree.h) keeps the previous burst's path in a second register set and switches between the two
    without a descent from memory.  The bench-like run (256 slots x 800 sims, real net, refills, late-game start positions)
    with the second set on and off (BETAONE_BURST_TWO_PATHS=0): every finished game identical -- moves, pi bits, z, terminal --
    and the switches do occur (profile counter 15)."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet
    from betaone_amd.rollout import Rollout

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, LIMIT, PREROLL, STEPS = 256, 800, 28, 40, 36
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_split").to("cuda:0")
    runs = {}
    for two in ("1", "0"):
        monkeypatch.setenv("BETAONE_BURST_TWO_PATHS", two)
        switches = []
        orig_close = Rollout.close

        def close_and_count(self, _sw=switches, _orig=orig_close):
            _sw.append(int(self.eng.profile(0)[:, 15].sum()))
            _orig(self)

        orig_init = Rollout.__init__

        def init_and_profile(self, *a, _orig=orig_init, **k):
            _orig(self, *a, **k)
            self.eng.profile(1, read=False)

        monkeypatch.setattr(Rollout, "close", close_and_count)
        monkeypatch.setattr(Rollout, "__init__", init_and_profile)
        fins, fens, _, _, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS, (), record=False)
        monkeypatch.setattr(Rollout, "close", orig_close)
        monkeypatch.setattr(Rollout, "__init__", orig_init)
        runs[two] = (fins, switches[0])
    (on, sw_on), (off, sw_off) = runs["1"], runs["0"]
    assert sw_on > 0 and sw_off == 0, (sw_on, sw_off)
    assert set(on) == set(off) and len(on) > 100
    for gid in on:
        a, b = on[gid], off[gid]
        assert list(a.moves) == list(b.moves) and a.terminal == b.terminal and a.outcome == b.outcome, gid
        assert len(a.pis) == len(b.pis)
        for (ia, va), (ib, vb) in zip(a.pis, b.pis):
            assert np.array_equal(ia, ib) and np.array_equal(np.asarray(va).view(np.uint32), np.asarray(vb).view(np.uint32)), gid


def test_bench_steady_state_path_with_refills_and_late_game_positions_matches_oracle(env):
    """The timed region of bench.py, oracle-checked at size: 256 slots x 800 sims x 8+2 x 128 on the Winograd tower, hipGraph
    on, native RNG, staggered starts, games ending (move limit 28, mates, claimed draws) and their slots refilled on the side
    stream while the others search, refilled slots' first searches begun without a host round trip, late-game start positions
    (terminal bursts and the 96-simulation yield at 800 sims with the real net).
      (1) the product path (n-iteration graphs) and the same run launched one iteration at a time finish the SAME games, bit
          for bit, for every game id;
      (2) every game that passed through a watched slot -- first occupants and games that entered through a refill alike --
          replayed through the CPU oracle from the recorded seam: moves, pi bits, z (with its sign), terminal code."""
    import torch
    import network
    from betaone_amd import engine as E
    from betaone_amd.fused_net import FusedPolicyValueNet
    from oracle import oracle as O

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, LIMIT, PREROLL, STEPS, WATCH = 256, 800, 28, 40, 36, (0, 5, 13, 101, 255)
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_split").to("cuda:0")  # (the route of nn_tune.kernel_route for this shape)
    prod, fens_p, _, slot_p, n_graphs = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS, WATCH, record=False)
    assert n_graphs > 0  # the product run did replay n-iteration graphs
    recd, fens_r, rec, slot_r, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS, WATCH, record=True)
    # (1) same games on both launch forms
    assert sorted(prod) == sorted(recd) and len(prod) >= G and fens_p == fens_r and slot_p == slot_r
    assert sum(1 for f in prod.values() if f.terminal != 0) >= 8          # games did end by the rules, not only by the move limit
    assert sum(1 for gid in prod if gid >= G) >= G // 2                     # games that entered through a refill finished too
    for gid, a in prod.items():
        b = recd[gid]
        assert a.moves == b.moves and a.terminal == b.terminal and a.outcome == b.outcome, gid
        assert len(a.pis) == len(b.pis) == len(a.moves) - a.first_ply, gid
        for (ia, va), (ib, vb) in zip(a.pis, b.pis):
            assert ia.tolist() == ib.tolist() and va.view(np.uint32).tolist() == vb.view(np.uint32).tolist(), gid
    # (2) the watched slots' games against the oracle
    watched = [gid for gid, sl in slot_r.items() if sl in WATCH]
    assert sum(1 for gid in watched if gid >= G) >= 4 and sum(1 for gid in watched if fens_r.get(gid)) >= 1
    late = [gid for gid in watched if fens_r.get(gid)]
    n_checked = 0
    for gid in watched:
        fin = recd[gid]
        ref = O.self_play(rec.eval_fn, np.random.RandomState(gid), O.default_config(num_simulations=SIMS, max_game_moves=LIMIT),
                          start_fen=fens_r.get(gid) or "")
        assert [O.move_to_uci(m) for m in ref["moves"]] == [E.move_to_uci(m) for m in fin.moves], gid
        term = 0 if ref["termination"] == 0 else (1 if ref["termination"] == 1 else 2)
        assert term == fin.terminal and ref["outcome"] == fin.outcome, gid
        assert len(ref["records"]) == len(fin.pis), gid
        for i, ((_, rpi, rz), (idx, val)) in enumerate(zip(ref["records"], fin.pis)):
            assert sorted(np.nonzero(rpi)[0].tolist()) == sorted(idx.tolist()), gid
            for j, v in zip(idx, val):
                assert np.float32(rpi[j]).view(np.uint32) == np.float32(v).view(np.uint32), gid
            z = fin.z(i)
            assert rz == z and np.signbit(rz) == np.signbit(z), gid
        n_checked += 1
    assert n_checked >= 8 and late, (n_checked, late)


@pytest.mark.parametrize("cohorts,cu_masks", [(1, None), (4, "contiguous")])
def test_step_kernel_finishing_the_evaluate_stage_plays_the_games_of_the_recorded_seam(env, cohorts, cu_masks, monkeypatch):
    """bench.py's default path since ABI 6: the evaluate stage ends behind bo_k_heads_tiles and bo_k_step finishes the row it consumes
    (Rollout.step_tail: policy_kind "logits" + FusedPolicyValueNet.forward_tail -> Engine.step_heads).  Its softmax and value are
    bo_k_heads_rows' operations in bo_k_heads_rows' order, so the bench-like steady-state run (256 slots x 800 sims, real 8+2 x 128 net,
    refills, late-game positions) must finish, game for game and bit for bit, the games of the run whose probabilities and values were
    written by bo_k_heads_rows -- the seam the test above records and replays through the CPU oracle.  Also with BETAONE_STEP_TAIL=0
    (logits, value from the rows kernel, softmax in the step kernel): the same games."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, LIMIT, PREROLL, STEPS = 256, 800, 28, 40, 36
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_split").to("cuda:0")
    seam, fens_s, _, _, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS, (), record=False)
    assert _LAST_RUN["step_tail"] is False
    tail, fens_t, _, _, n_graphs = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS + (1 if cohorts > 1 else 0), (), record=False, cohorts=cohorts,
                                                   cu_masks=cu_masks, policy_kind="logits")
    assert _LAST_RUN["step_tail"] is True and n_graphs > 0
    runs = [(tail, fens_t)]
    if cohorts == 1:
        monkeypatch.setenv("BETAONE_STEP_TAIL", "0")
        mid, fens_m, _, _, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS, (), record=False, policy_kind="logits")
        assert _LAST_RUN["step_tail"] is False
        runs.append((mid, fens_m))
    for other, fens_o in runs:
        # (cohorts refill slots in another order: ids handed out in the same ply can meet different start positions -- compare the rest)
        common = [gid for gid in seam if gid in other and fens_s.get(gid) == fens_o.get(gid)]
        assert len(common) >= G and sum(1 for gid in common if gid >= G) >= G // 4 and (cohorts > 1 or sorted(seam) == sorted(other))
        assert sum(1 for gid in common if seam[gid].terminal != 0) >= 4
        for gid in common:
            a, b = seam[gid], other[gid]
            assert list(a.moves) == list(b.moves) and a.terminal == b.terminal and a.outcome == b.outcome, gid
            assert len(a.pis) == len(b.pis), gid
            for (ia, va), (ib, vb) in zip(a.pis, b.pis):
                assert ia.tolist() == ib.tolist() and va.view(np.uint32).tolist() == vb.view(np.uint32).tolist(), gid


@pytest.mark.parametrize("cohorts,cu_masks", [(2, None), (4, None), (4, "contiguous")])
def test_cohorts_on_their_own_streams_finish_the_games_of_the_single_rollout(env, cohorts, cu_masks):
    """rollout.CohortRollout (bench.py --cohorts K): the 256 slots as K cohorts, each with its own engine, HIP stream and captured
    graphs, the plies software-pipelined (ply_end of one ply followed at once by ply_begin of the next, cohort after cohort).
    Games are independent (main.py:160-175), so the bench-like steady-state run -- refills on the side streams, late-game start
    positions, game ids handed out in the order slots fall free -- must finish the very games the single Rollout finishes (which
    the test above replays through the oracle): moves, pi bits, z, terminal code, for every game id both runs finished."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, LIMIT, PREROLL, STEPS = 256, 800, 28, 40, 36
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_split").to("cuda:0")
    one, fens1, _, _, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS, (), record=False)
    # cu_masks: every cohort's stream confined to its own quarter of the CUs (bo_stream_create_cu_mask) -- placement only, same games
    many, fensk, _, slots, n_graphs = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS + 1, (), record=False, cohorts=cohorts, cu_masks=cu_masks)
    assert n_graphs >= cohorts and set(slots.values()) == set(range(G))   # every cohort replayed graphs; finished games carry global slots
    # a game's id, seed and start position are handed out when a slot falls free: the ORDER in which slots are refilled differs
    # between the two schedules only among games that end in the same ply, so ids can be attached to different start positions
    # there -- compare every id whose start position is the same in both runs (all first occupants, and nearly all refills)
    both = [gid for gid in one if gid in many and fens1.get(gid) == fensk.get(gid)]
    assert len(both) >= G and sum(1 for gid in both if gid >= G) >= G // 4
    assert sum(1 for gid in both if one[gid].terminal != 0) >= 4
    for gid in both:
        a, b = one[gid], many[gid]
        assert list(a.moves) == list(b.moves) and a.terminal == b.terminal and a.outcome == b.outcome, gid
        assert len(a.pis) == len(b.pis) == len(a.moves) - a.first_ply, gid
        for (ia, va), (ib, vb) in zip(a.pis, b.pis):
            assert ia.tolist() == ib.tolist() and va.view(np.uint32).tolist() == vb.view(np.uint32).tolist(), gid


@pytest.mark.parametrize("cohorts", [1, 4])
def test_device_turn_finishes_the_games_of_the_host_turn_on_the_bench_like_run(env, cohorts, monkeypatch):
    """bo_selfplay_autoturn (the ply's turn as two kernels behind the searches: csrc/bo_tree.h bo_k_turn_sample / bo_k_turn_play) against
    the host-made turn (bo_selfplay_turn; BETAONE_DEVICE_TURN=0), which the oracle replay above pins: the bench-like run -- 256 slots x
    800 sims, the real net on the split-precision tower, captured graphs, refills on the side streams, late-game start positions whose
    games cross the temperature threshold (fullmove 30: p ** 10 through the host-built table) and end by rule -- finishes the same games
    either way: moves, pi bits, z, terminal code.  As one Rollout and as four cohorts on CU-masked streams."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet
    from betaone_amd.rollout import Rollout

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, LIMIT, PREROLL, STEPS = 256, 800, 28, 40, 36
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_split").to("cuda:0")
    seen = []
    orig = Rollout._enqueue_autoturn

    def counting(self, *a, **kw):
        seen.append(1)
        return orig(self, *a, **kw)

    monkeypatch.setattr(Rollout, "_enqueue_autoturn", counting)
    monkeypatch.setenv("BETAONE_DEVICE_TURN", "0")
    host, fens_h, _, _, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS + (1 if cohorts > 1 else 0), (), record=False, cohorts=cohorts)
    assert not seen
    monkeypatch.setenv("BETAONE_DEVICE_TURN", "1")
    dev, fens_d, _, _, _ = _bench_like_run(net, G, SIMS, LIMIT, PREROLL, STEPS + (1 if cohorts > 1 else 0), (), record=False, cohorts=cohorts)
    assert len(seen) >= STEPS * cohorts  # (the device turn is what ran)
    both = [gid for gid in host if gid in dev and fens_h.get(gid) == fens_d.get(gid)]
    assert len(both) >= G and sum(1 for gid in both if host[gid].terminal != 0) >= 4
    for gid in both:
        a, b = host[gid], dev[gid]
        assert list(a.moves) == list(b.moves) and a.terminal == b.terminal and a.outcome == b.outcome, gid
        assert len(a.pis) == len(b.pis), gid
        for (ia, va), (ib, vb) in zip(a.pis, b.pis):
            assert ia.tolist() == ib.tolist() and va.view(np.uint32).tolist() == vb.view(np.uint32).tolist(), gid


def test_iterations_in_one_graph_launch_play_the_same_games(env):
    """Rollout._eval_and_step_n: a search's expected evaluations replayed as ONE graph of n iterations give bit-identical
    games (moves, pis) to n launches of the one-iteration graph."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet
    from betaone_amd.rollout import Rollout

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, PLIES = 64, 400, 6
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_wg").to("cuda:0")
    games = []
    for max_iter in (1, Rollout.MAX_GRAPH_ITERATIONS):
        ro = Rollout(net, G, num_simulations=SIMS, mcts_batch_size=96, device="cuda:0", use_graph=True, rng_mode="native", policy_kind="probs")
        ro.MAX_GRAPH_ITERATIONS = max_iter
        if max_iter == 1:
            ro._eval_and_step_n = lambda n, ro=ro: [ro._eval_and_step() for _ in range(n)]
        ro.start_games(list(range(G)), list(range(G)), list(range(G)))
        for _ in range(PLIES):
            assert ro.play_ply() == G
        ro.eng.check_status()
        assert (len(ro._graphs_n) > 0) == (max_iter > 1)
        games.append([ro._finish(g, 0) for g in range(G)])
        ro.close()
    for a, b in zip(*games):
        assert a.moves == b.moves and len(a.pis) == len(b.pis) == PLIES
        for (ia, va), (ib, vb) in zip(a.pis, b.pis):
            assert ia.tolist() == ib.tolist() and va.view(np.uint32).tolist() == vb.view(np.uint32).tolist()


def test_config2_shard_256_games_800_sims_10x128_graph_and_winograd_tower(env):
    """configs[2]'s per-GPU shard exactly as bench.py runs it: 256 games x 800 sims, net 8+2 x 128 fp32 on the hand-written
    Winograd tower kernel, native RNG streams, hipGraph-captured evaluate -> step."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet
    from betaone_amd.rollout import Rollout

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 8, 2, 128
    G, SIMS, PLIES, WATCH = 256, 800, 4, (0, 101, 255)
    torch.manual_seed(0)
    net = FusedPolicyValueNet(network.PolicyValueNet().to("cuda:0").eval(), conv="tower_split").to("cuda:0")  # (the route of nn_tune.kernel_route for this shape)
    ro = Rollout(net, G, num_simulations=SIMS, mcts_batch_size=96, device="cuda:0", use_graph=True, rng_mode="native",
                 policy_kind="probs")
    rec = SeamRecorder(ro, WATCH)
    ro.start_games(list(range(G)), list(range(G)), list(range(G)))
    for _ in range(PLIES):
        assert ro.play_ply() == G
    assert ro._graph is not None and ro.n_forward >= PLIES * 10
    _check_games_against_oracle(ro, rec, WATCH, SIMS, PLIES)
    ro.close()


def test_config4_shard_512_games_800_sims_20x256_fp16_tower(env):
    """configs[4]'s per-GPU shard: 512 games x 800 sims, net 15+5 x 256 evaluated in fp16 by the two-boards-per-workgroup
    tower kernel; the tree is float32 given the (fp16-computed) seam, so it must still equal the oracle's bit for bit.
    Plus: the fp16 logits stay within the stated bound of the same net in float32."""
    import torch
    import network
    from betaone_amd.fused_net import FusedPolicyValueNet
    from betaone_amd.rollout import Rollout

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
    G, SIMS, PLIES, WATCH = 512, 800, 3, (0, 255, 511)
    torch.manual_seed(0)
    plain = network.PolicyValueNet().to("cuda:0").eval()
    net = FusedPolicyValueNet(plain, conv="tower_f16").to("cuda:0")
    ro = Rollout(net, G, num_simulations=SIMS, mcts_batch_size=96, device="cuda:0", use_graph=True, rng_mode="native",
                 policy_kind="probs")
    rec = SeamRecorder(ro, WATCH)
    ro.start_games(list(range(G)), list(range(G)), list(range(G)))
    for _ in range(PLIES):
        assert ro.play_ply() == G
    _check_games_against_oracle(ro, rec, WATCH, SIMS, PLIES)
    # fp16 evaluate stage vs the float32 net on the 512 positions now in the NN rows: error of the size of torch-fp16's own
    x = ro.nn_in.clone()
    half = plain.for_inference(dtype=torch.float16, channels_last=False)
    with torch.no_grad():
        l32, v32 = plain(x)
        l16, v16 = net(x)
        lt, vt = half(x.half())
    e_ours = max((l32 - l16.float()).abs().max().item(), (v32 - v16.float()).abs().max().item())
    e_torch = max((l32 - lt.float()).abs().max().item(), (v32 - vt.float()).abs().max().item())
    scale = max(1.0, l32.abs().max().item())
    assert e_ours <= max(3.0 * e_torch, 4e-3 * scale), (e_ours, e_torch, scale)
    ro.close()


def test_config3_uci_search_1600_sims_20x256_dropin_graph_path(env):
    """configs[3]: the call uci.py makes (uci.py:62-63) -- mcts.run_mcts at 1600 simulations with the 15+5 x 256 net at batch 1
    through the captured-graph path of the drop-in, history in uci.py's form (root board duplicated, quirk E5).  Best move, pi
    and the whole tree against oracle.run_mcts fed with the recorded seam."""
    import torch
    import chess
    import mcts
    import network
    import utils
    from engine_harness import canonical_tree
    from fake_model import hash_init_, planes_key
    from oracle import oracle as O

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.POLICY_SOFTMAX = 1600, 96, "torch"
    model = hash_init_(network.PolicyValueNet().eval()).to("cuda")
    moves = "e2e4 e7e5 g1f3 b8c6 f1b5 a7a6 b5a4 g8f6".split()
    board = chess.Board()
    tracker = utils.RepetitionTracker()
    tracker.add_board(board)
    hist = [board.copy()]
    for u in moves:
        board.push(chess.Move.from_uci(u)); tracker.add_board(board); hist.append(board.copy())
    history = hist[-8:][-7:]  # uci.py:62: history[-7:] of a list that already ends with the root board

    seam = {}
    orig = mcts._GraphStep.__call__

    def recording_call(self):
        planes = self.nn_in.cpu().numpy()
        orig(self)
        seam[planes_key(planes[0])] = (self.out[0][0].float().cpu().numpy().copy(), np.float32(self.out[1].reshape(-1)[0].item()))

    mcts._GraphStep.__call__ = recording_call
    try:
        results = []
        for _ in range(2):  # the first call tunes + captures, the second replays
            np.random.seed(9)
            best, pi = mcts.run_mcts(board, model, history, tracker)
            results.append((best.uci(), pi.copy()))
        eng = next(iter(mcts._ctx.values()))[0]
        got = canonical_tree(eng.debug_tree(0))
        graph_step = next(iter(mcts._fast.values()))[2]
        assert graph_step.kind == 2 and graph_step.graph is not None  # BO_POLICY_PROBS through a captured hipGraph
    finally:
        mcts._GraphStep.__call__ = orig
    assert results[0][0] == results[1][0] and np.array_equal(results[0][1], results[1][1])

    def eval_fn(planes):
        probs = np.zeros((planes.shape[0], 4672), np.float32)
        vals = np.zeros(planes.shape[0], np.float32)
        for i in range(planes.shape[0]):
            probs[i], vals[i] = seam[planes_key(planes[i])]
        return probs, vals

    ob = O.Board()
    ot = O.PyTracker(); ot.add_board(ob)
    for u in moves:
        ob.push(u); ot.add_board(ob)
    oh = ob.positions()[-8:][-7:]
    r = O.run_mcts(ob, oh, ot, eval_fn, np.random.RandomState(9), O.default_config(num_simulations=1600, batch_size=96))
    assert results[0][0] == O.move_to_uci(r["best"])
    assert np.array_equal(results[0][1].view(np.uint32), r["pi"].view(np.uint32))
    exp = {"/".join(k): list(v) for k, v in O.canonical_tree(r["nodes"]).items()}
    assert got == exp
    assert len(seam) >= 1 + 1600 // 96  # one root evaluation + one leaf per batch


def _uci_position(moves):
    """What uci.py builds for `position startpos moves ...` (uci.py:161-199): board, tracker of every position, history[-8:]
    INCLUDING the current board."""
    import chess
    import utils

    board = chess.Board()
    tracker = utils.RepetitionTracker()
    tracker.add_board(board)
    hist = [board.copy()]
    for u in moves:
        board.push(chess.Move.from_uci(u)); tracker.add_board(board); hist.append(board.copy())
    return board, hist[-8:], tracker


def test_config3_interrupted_search_on_gpu_equals_oracle_at_the_simulations_done(env):
    """Row f2 on the product path: a 1600-simulation search (15+5 x 256 net, captured graph) interrupted after k replays
    returns the reference's result for NUM_SIMULATIONS = k x 96, checked against the oracle from the recorded seam."""
    import mcts
    import network
    from fake_model import hash_init_, planes_key
    from oracle import oracle as O

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.POLICY_SOFTMAX = 1600, 96, "torch"
    model = hash_init_(network.PolicyValueNet().eval()).to("cuda")
    moves = "d2d4 d7d5 c2c4 e7e6 b1c3".split()
    board, hist8, tracker = _uci_position(moves)
    history = hist8[-7:]
    np.random.seed(4)
    mcts.run_mcts(board, model, history, tracker)  # tune + capture
    assert mcts.last_search == {"simulations": 1600, "stopped": False}

    seam, calls = {}, [0]
    orig = mcts._GraphStep.__call__

    def recording_call(self):
        planes = self.nn_in.cpu().numpy()
        orig(self)
        seam[planes_key(planes[0])] = (self.out[0][0].float().cpu().numpy().copy(), np.float32(self.out[1].reshape(-1)[0].item()))
        calls[0] += 1
        if calls[0] == 8:
            mcts.request_stop()

    mcts._GraphStep.__call__ = recording_call
    try:
        np.random.seed(4)
        best, pi = mcts.run_mcts(board, model, history, tracker)
    finally:
        mcts._GraphStep.__call__ = orig
        mcts.stop_event.clear()
    assert mcts.last_search == {"simulations": 7 * 96, "stopped": True}   # root evaluation + 7 leaf evaluations consumed

    def eval_fn(planes):
        probs = np.zeros((planes.shape[0], 4672), np.float32)
        vals = np.zeros(planes.shape[0], np.float32)
        for i in range(planes.shape[0]):
            probs[i], vals[i] = seam[planes_key(planes[i])]
        return probs, vals

    ob = O.Board()
    ot = O.PyTracker(); ot.add_board(ob)
    for u in moves:
        ob.push(u); ot.add_board(ob)
    r = O.run_mcts(ob, ob.positions()[-8:][-7:], ot, eval_fn, np.random.RandomState(4), O.default_config(num_simulations=7 * 96))
    assert best.uci() == O.move_to_uci(r["best"])
    assert np.array_equal(pi.view(np.uint32), r["pi"].view(np.uint32))


def test_config3_uci_shaped_session_latency_and_stop(env):
    """The call sequence of uci.py on the GPU box (the reference file itself cannot travel): `position ... / go movetime /
    stop` -- a worker thread runs run_mcts back to back on a copy of the position (uci.py:48-120) while the main thread waits,
    then sets the stop flag.  Config 3: 1600 simulations, 15+5 x 256 net, hipGraph path.  Asserts the per-search latency
    (<= 12 ms median) and that a stop ends the running search between two steps."""
    import threading
    import time

    import torch
    import mcts
    import network

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 15, 5, 256
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE = 1600, 96
    torch.manual_seed(0)
    model = network.PolicyValueNet().to("cuda").eval()
    board, history, tracker = _uci_position("e2e4 e7e5 g1f3 b8c6 f1b5 a7a6".split())
    out = {"lat": [], "moves": [], "stopped_mid_search": 0}

    def search_worker(board, history, tracker, time_limit_ms):   # the shape of uci.py:48-120
        start = time.time()
        while True:
            t0 = time.perf_counter()
            best, _pi = mcts.run_mcts(board, model, history[max(0, len(history) - 7):], tracker)
            out["lat"].append((time.perf_counter() - t0) * 1e3)
            out["moves"].append(best.uci())
            out["stopped_mid_search"] += 1 if mcts.last_search["stopped"] else 0
            if mcts.stop_event.is_set() or (time.time() - start) * 1e3 >= time_limit_ms:
                break

    np.random.seed(0)
    for _ in range(3):
        mcts.run_mcts(board, model, history[-7:], tracker)   # first call tunes the inference copy and captures the graph
    # go movetime 400
    th = threading.Thread(target=search_worker, args=(board.copy(), list(history), tracker, 400.0), daemon=True)
    th.start(); th.join(timeout=30.0)
    assert not th.is_alive() and len(out["lat"]) >= 10
    med = float(np.median(out["lat"]))
    print(f"uci-shaped session: {len(out['lat'])} searches of 1600 sims in 400 ms, median {med:.2f} ms, min {min(out['lat']):.2f} ms")
    assert med <= 12.0, out["lat"]
    legal = {m.uci() for m in board.legal_moves}
    assert set(out["moves"]) <= legal
    # go infinite ... stop
    config.NUM_SIMULATIONS = 96 * 4000   # a long search (~2 s): the stop must land inside it
    mcts.request_stop()
    mcts.run_mcts(board, model, history[-7:], tracker)   # builds the engine + graph for these settings, returns at once
    mcts.stop_event.clear()
    assert mcts.last_search["stopped"]
    n_before = len(out["lat"])
    th = threading.Thread(target=search_worker, args=(board.copy(), list(history), tracker, 60_000.0), daemon=True)
    th.start()
    time.sleep(0.3)
    t_stop = time.perf_counter()
    mcts.request_stop()
    th.join(timeout=10.0)
    dt_stop = (time.perf_counter() - t_stop) * 1e3
    mcts.stop_event.clear()
    assert not th.is_alive()
    assert out["stopped_mid_search"] == 1 and len(out["lat"]) == n_before + 1
    assert 0 < mcts.last_search["simulations"] < 96 * 4000 and mcts.last_search["simulations"] % 96 == 0
    assert out["moves"][-1] in legal
    print(f"stop -> bestmove after {dt_stop:.1f} ms ({mcts.last_search['simulations']} simulations done)")
    assert dt_stop < 250.0


def test_orchestrator_iteration_on_gpu_writes_reference_pickles_and_resumes(env, tmp_path):
    """Row f4 on the product path: betaone_amd.selfplay_main.run_iteration (the replacement of main.py:142-191) plays 24 games
    in 8 slots on cuda:0, writes the reference's pickles, resumes by skipping what is on disk (main.py:26-36), and what it
    wrote loads through the contract of train.load_recent_data / ChessDataset.__getitem__ (train.py:179-184, :207-214)."""
    import glob
    import pickle

    import torch
    import network
    from betaone_amd import selfplay_main as M

    config = env
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 3, 1, 64
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 100, 96, 12
    config.DATA_DIR = str(tmp_path / "data")
    torch.manual_seed(0)
    model = network.PolicyValueNet().to("cuda").eval()
    logs = []
    done = M.run_iteration(model, 5, n_games=24, n_slots=8, log=logs.append)
    assert sorted(done) == list(range(24)) and all(1 <= v <= 12 for v in done.values())
    files = sorted(glob.glob(os.path.join(config.DATA_DIR, "iter_5", "game_*.pkl")))
    assert len(files) == 24
    all_data = []
    for f in files:  # train.load_recent_data's acceptance test (train.py:207-214)
        game = pickle.load(open(f, "rb"))
        assert isinstance(game, list) and game
        all_data.extend(game)
    assert len(all_data) == sum(done.values())
    for state, policy, value in all_data[:50]:  # ChessDataset.__getitem__ (train.py:179-184)
        policy_tensor = torch.from_numpy(policy).float()
        value_tensor = torch.tensor([value], dtype=torch.float32)
        assert isinstance(state, torch.Tensor) and state.dtype == torch.float32 and tuple(state.shape) == (120, 8, 8)
        assert state.device.type == "cpu" and tuple(policy_tensor.shape) == (4672,) and tuple(value_tensor.shape) == (1,)
        assert abs(float(policy_tensor.sum()) - 1.0) < 1e-6 and float(value_tensor) in (-1.0, 0.0, 1.0)
    batch = torch.stack([d[0] for d in all_data[:16]])  # a training batch forms
    assert tuple(batch.shape) == (16, 120, 8, 8)
    # resume: remove three files, a second call plays exactly those games again and reproduces them (seeds travel with the ids)
    keep = {j: pickle.load(open(os.path.join(config.DATA_DIR, "iter_5", f"game_{j}.pkl"), "rb")) for j in (3, 11, 17)}
    for j in keep:
        os.remove(os.path.join(config.DATA_DIR, "iter_5", f"game_{j}.pkl"))
    assert M.pending_game_ids(config.DATA_DIR, 5, 24) == [3, 11, 17]
    again = M.run_iteration(model, 5, n_games=24, n_slots=8, log=logs.append)
    assert sorted(again) == [3, 11, 17]
    for j, old in keep.items():
        new = pickle.load(open(os.path.join(config.DATA_DIR, "iter_5", f"game_{j}.pkl"), "rb"))
        assert len(new) == len(old)
        assert all(torch.equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(new, old))
    assert M.run_iteration(model, 5, n_games=24, n_slots=8, log=logs.append) == {}


def test_bench_starts_its_own_ranks_and_exchanges_records(env, tmp_path):
    """`python bench.py --gpus 2` (no torchrun): the parent starts two ranks before touching a GPU and relays rank 0's JSON
    line; both ranks play their own games (ids by rank), finished games travel through the pipelined exchange.  Two ranks
    share this box's one GPU here (gloo), as a rehearsal of the one-rank-per-GPU RCCL run."""
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--games", "16",
           "--sims", "60", "--net", "4x64", "--steps", "6", "--warmup", "1", "--preroll", "12", "--max-game-moves", "10",
           "--exchange-every", "2", "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric')]
    assert len(lines) == 1, p.stdout[-2000:]           # exactly one JSON line: rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0
    ex = d["record_exchange"]
    assert ex["records_received_rank0"] == d["games_finished_since_start"] > 0   # every rank's finished games arrived at rank 0
    assert ex["payload_gathers"] >= 1 and ex["size_gathers"] >= ex["payload_gathers"]


def test_record_exchange_through_rccl_on_one_rank(env):
    """records.PeriodicGameExchange on the RCCL backend (world size 1 on this box): staging buffers, side stream and the
    completion events of the asynchronous path; every record comes back, no tick blocks."""
    import torch
    import torch.distributed as dist
    from betaone_amd import engine as E, records
    from betaone_amd.rollout import FinishedGame, SparsePis

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        ex = records.PeriodicGameExchange(torch.device("cuda:0"), every=2)
        pos = [E.BoPosition() for _ in range(4)]
        got, sent = [], 0
        for step in range(9):
            fins = []
            if step % 3 == 0:
                pis = SparsePis(np.array([1, 2, 1], np.int32), np.array([[5, 0], [7, 9], [3, 0]], np.int32),
                                np.array([[1.0, 0.0], [0.25, 0.75], [1.0, 0.0]], np.float32))
                fins.append(FinishedGame(game_id=100 + step, slot=0, moves=[796, 3364, 100], positions=pos, pis=pis, outcome=0.0, terminal=2))
                sent += 1
            got.extend(ex.push(fins))
            torch.cuda.synchronize()
        got.extend(ex.flush())
        assert sorted(g["game_id"] for g in got) == [100, 103, 106] and sent == 3
        assert got[0]["pis"][1][0].tolist() == [7, 9] and got[0]["pis"][1][1].tolist() == [0.25, 0.75]
        assert ex.blocked_ticks == 0 and ex.n_size_gathers == 5 and 1 <= ex.n_payload_gathers <= 4
    finally:
        dist.destroy_process_group()
