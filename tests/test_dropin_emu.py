"""CPU tests of the drop-in surface (betaone_amd/dropin: config, network, utils, mcts, self_play) --
the host logic that marshals the caller's python-chess objects into the engine.  The engine behind it
is the wave-emulator build (test infrastructure); boards are oracle/shim `chess` boards standing in
for python-chess.  Bit-exact comparisons use a net with constant logits: softmax of equal logits is
exactly 1/4672 in any implementation, so the (priors, value) seam is identical on both sides."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "shim"))

import chess  # noqa: E402  (oracle/shim)
from betaone_amd import dropin  # noqa: E402

dropin.install()
import config  # noqa: E402
import mcts  # noqa: E402
import network  # noqa: E402
import self_play  # noqa: E402
import utils  # noqa: E402

import golden_util as G  # noqa: E402
from engine_harness import emulator_backend  # noqa: E402
from fake_model import FakeNet, fake_logits_values, hash_init_  # noqa: E402
from oracle import oracle as O  # noqa: E402


@pytest.fixture(autouse=True)
def _emu_backend(request):
    """Every test of this file drives the drop-in modules on the wave-emulator build of the device code (patched in from
    the outside, see engine_harness.emulator_backend) -- except those marked `product_backend`."""
    saved = {k: getattr(config, k) for k in ("NUM_SIMULATIONS", "MCTS_BATCH_SIZE", "MAX_GAME_MOVES", "DIRICHLET_ALPHA", "DATA_DIR", "COHORTS")}
    mcts._ctx.clear()
    if request.node.get_closest_marker("product_backend"):
        yield
    else:
        with emulator_backend():
            yield
    mcts._ctx.clear()
    for k, v in saved.items():
        setattr(config, k, v)


def uniform_eval(salt):
    def fn(planes):
        _, v = fake_logits_values(planes, 0.0, salt)
        return np.full((planes.shape[0], 4672), np.float32(1.0) / np.float32(4672.0), dtype=np.float32), v
    return fn


def context(fen, moves, uci_style=False):
    board = chess.Board(fen)
    tracker = utils.RepetitionTracker()
    tracker.add_board(board)
    hist = [board.copy()]
    for u in moves:
        board.push(chess.Move.from_uci(u))
        tracker.add_board(board)
        hist.append(board.copy())
    history = hist[-8:][-7:] if uci_style else hist[max(0, len(hist) - 8):-1]
    return board, history, tracker


def oracle_context(fen, moves, uci_style=False):
    b = O.Board(fen)
    trk = O.PyTracker()
    trk.add_board(b)
    for u in moves:
        b.push(u)
        trk.add_board(b)
    pos = b.positions()
    return b, (pos[-8:][-7:] if uci_style else pos[max(0, len(pos) - 8):-1]), trk


CASES = [
    (chess.STARTING_FEN, [], False, 120),
    (chess.STARTING_FEN, "e2e4 e7e5 g1f3 b8c6 f1b5 a7a6 b5a4 g8f6 e1g1".split(), False, 200),
    (chess.STARTING_FEN, "e2e4 c7c5 g1f3".split(), True, 100),
    ("rnbqkbnr/ppp1pppp/8/8/3pP3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 3", [], False, 100),
    (chess.STARTING_FEN, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6".split(), False, 150),
]


@pytest.mark.parametrize("fen,moves,uci_style,sims", CASES)
def test_run_mcts_dropin_matches_oracle(fen, moves, uci_style, sims):
    config.NUM_SIMULATIONS = sims
    board, history, tracker = context(fen, moves, uci_style)
    np.random.seed(7)
    best, pi = mcts.run_mcts(board, FakeNet(scale=0.0, salt=3), history, tracker)
    ob, oh, ot = oracle_context(fen, moves, uci_style)
    r = O.run_mcts(ob, oh, ot, uniform_eval(3), np.random.RandomState(7), O.default_config(num_simulations=sims))
    assert isinstance(best, chess.Move) and best.uci() == O.move_to_uci(r["best"])
    assert pi.dtype == np.float32 and pi.shape == (4672,)
    assert np.array_equal(pi.view(np.uint32), r["pi"].view(np.uint32))
    assert board.move_stack == [chess.Move.from_uci(u) for u in moves]  # caller's board untouched (mcts.py:36)


def test_run_mcts_raises_on_root_without_moves():
    board, history, tracker = context("k6R/8/1K6/8/8/8/8/8 b - - 1 1", [])
    with pytest.raises(ValueError):
        mcts.run_mcts(board, FakeNet(scale=0.0), history, tracker)
    assert mcts.MCTS(FakeNet(scale=0.0)).search  # north-star alias exists


def test_run_self_play_game_dropin_matches_oracle(tmp_path):
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 40, 16, 7
    np.random.seed(11)
    data = self_play.run_self_play_game(FakeNet(scale=0.0, salt=5), 42)
    ref = O.self_play(uniform_eval(5), np.random.RandomState(11),
                      O.default_config(num_simulations=40, batch_size=16, max_game_moves=7))
    assert isinstance(data, list) and len(data) == len(ref["records"]) == 7
    for (st, pi, z), (rst, rpi, rz) in zip(data, ref["records"]):
        assert isinstance(st, torch.Tensor) and st.dtype == torch.float32 and tuple(st.shape) == (120, 8, 8)
        assert np.array_equal(st.numpy(), rst)
        assert isinstance(pi, np.ndarray) and np.array_equal(pi.view(np.uint32), rpi.view(np.uint32))
        assert isinstance(z, float) and z == rz and np.signbit(z) == np.signbit(rz)
    # save_game_data keeps the reference's pickle layout (train.py:207-214 expects a non-empty list)
    config.DATA_DIR = str(tmp_path)
    self_play.save_game_data(data, 3, 42)
    import pickle

    back = pickle.load(open(tmp_path / "iter_3" / "game_42.pkl", "rb"))
    assert isinstance(back, list) and len(back) == 7 and torch.equal(back[0][0], data[0][0])
    assert self_play.play_game is self_play.run_self_play_game


def test_batched_games_are_independent_of_slot_count():
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 5
    model = FakeNet(scale=0.0, salt=9)
    a = self_play.run_self_play_games(model, [0, 1, 2, 3, 4], seeds=[10, 11, 12, 13, 14], n_slots=5)
    b = self_play.run_self_play_games(model, [0, 1, 2, 3, 4], seeds=[10, 11, 12, 13, 14], n_slots=2)
    for gid in range(5):
        assert len(a[gid]) == len(b[gid]) == 5
        for (s1, p1, z1), (s2, p2, z2) in zip(a[gid], b[gid]):
            assert torch.equal(s1, s2) and np.array_equal(p1, p2) and z1 == z2
    ref = O.self_play(uniform_eval(9), np.random.RandomState(12),
                      O.default_config(num_simulations=30, batch_size=16, max_game_moves=5))
    for (s1, p1, z1), (rs, rp, rz) in zip(a[2], ref["records"]):
        assert np.array_equal(s1.numpy(), rs) and np.array_equal(p1, rp) and z1 == rz


def test_utils_codec_and_encoding_match_oracle():
    for e in G.load_codec():
        b = chess.Board(e["fen"])
        assert utils.test_move_indexing(b) == 0
        assert [[m.uci(), utils.move_to_index(m)] for m in b.legal_moves] == e["moves"]
        for uci, idx in e["moves"]:
            assert utils.index_to_move(idx, b).uci() == uci
    with pytest.raises(ValueError):
        utils.index_to_move(4672, chess.Board())
    with pytest.raises(ValueError):
        utils.index_to_move(0 * 73 + 64, chess.Board())  # under-promotion plane without a pawn on a1
    for fen, moves, uci_style, _ in CASES:
        board, history, tracker = context(fen, moves, uci_style)
        enc = utils.encode_board(board, (history + [board])[-8:], tracker)
        ob, oh, ot = oracle_context(fen, moves, uci_style)
        assert np.array_equal(enc.numpy(), O.encode_board((list(oh) + [ob.pos])[-8:], ot))
    t = utils.RepetitionTracker()
    b = chess.Board()
    assert t.repetitions(b) == 0
    t.add_board(b); t.add_board(b); t.add_board(b)
    assert t.repetitions(b) == 2 and t.get_count(b) == 3
    assert utils.get_game_outcome(chess.Board()) is None
    assert utils.get_game_outcome(chess.Board("k6R/8/1K6/8/8/8/8/8 b - - 1 1")) == 1.0
    assert utils.get_game_outcome(chess.Board("8/8/8/8/8/2k5/8/K7 w - - 0 1")) == 0.0


NET_SIZES = {"3+1x64": (3, 1, 64), "8+2x128": (8, 2, 128), "15+5x256": (15, 5, 256)}


@pytest.mark.parametrize("name", list(NET_SIZES))
def test_network_matches_reference_outputs(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "g1_net.npz"))
    saved = (config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS)
    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = NET_SIZES[name]
    try:
        net = hash_init_(network.PolicyValueNet().eval())
        assert len(net.state_dict()) == int(z[f"nkeys_{name}"])
        assert sum(p.numel() for p in net.parameters()) == int(z[f"nparams_{name}"])
        x = torch.from_numpy(z["inputs"])
        with torch.no_grad():
            logits, value = net(x)
            flogits, fvalue = net.for_inference(channels_last=False)(x)
        assert tuple(logits.shape) == (3, 4672) and tuple(value.shape) == (3, 1)
        assert np.abs(logits.numpy() - z[f"logits_{name}"]).max() < 1e-4   # north_star tolerance
        assert np.abs(value.numpy() - z[f"value_{name}"]).max() < 1e-4
        assert np.abs(flogits.numpy() - z[f"logits_{name}"]).max() < 1e-4  # BN-folded inference copy
        assert np.abs(fvalue.numpy() - z[f"value_{name}"]).max() < 1e-4
    finally:
        config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = saved


@pytest.mark.product_backend
def test_product_paths_refuse_to_run_without_the_gpu():
    from betaone_amd import engine as E

    if not torch.cuda.is_available():
        board, history, tracker = context(chess.STARTING_FEN, [])
        with pytest.raises(E.EngineError):
            mcts.run_mcts(board, FakeNet(scale=0.0), history, tracker)
        with pytest.raises(E.EngineError):
            self_play.run_self_play_game(FakeNet(scale=0.0), 0)


def test_selfplay_orchestration_writes_reference_pickles_and_resumes(tmp_path):
    """betaone_amd/selfplay_main.py (row f4): one iteration's games -> DATA_DIR/iter_i/game_j.pkl in the reference's
    format (train.py:196-217 expects a non-empty list per file); a second call skips what is on disk (main.py:26-36)."""
    import pickle
    from betaone_amd import selfplay_main as M

    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 4
    config.DATA_DIR = str(tmp_path / "data")
    model = FakeNet(scale=0.0, salt=2)
    logs = []
    done = M.run_iteration(model, 7, n_games=5, n_slots=3, log=logs.append)
    assert sorted(done) == [0, 1, 2, 3, 4] and all(v == 4 for v in done.values())
    for j in range(5):
        data = pickle.load(open(tmp_path / "data" / "iter_7" / f"game_{j}.pkl", "rb"))
        assert isinstance(data, list) and len(data) == 4
        st, pi, z = data[0]
        assert isinstance(st, torch.Tensor) and tuple(st.shape) == (120, 8, 8) and pi.shape == (4672,) and isinstance(z, float)
    os.remove(tmp_path / "data" / "iter_7" / "game_3.pkl")
    assert M.pending_game_ids(config.DATA_DIR, 7, 5) == [3]
    again = M.run_iteration(model, 7, n_games=5, n_slots=3, log=logs.append)
    assert sorted(again) == [3]
    # sharding: rank 1 of 2 plays the odd ids only, with the same per-game seeds
    config.DATA_DIR = str(tmp_path / "data2")
    odd = M.run_iteration(model, 7, n_games=5, n_slots=3, rank=1, world=2, log=logs.append)
    assert sorted(odd) == [1, 3]
    a = pickle.load(open(tmp_path / "data" / "iter_7" / "game_1.pkl", "rb"))
    b = pickle.load(open(tmp_path / "data2" / "iter_7" / "game_1.pkl", "rb"))
    assert all(torch.equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))


def test_compact_records_on_disk_yield_what_the_reference_pickles_yield(tmp_path):
    """Row f3 on disk: selfplay_main --records both writes the reference's pickles AND ~100 B/ply compact records
    (games_rank0.bog); records.CompactDataset over the compact file yields, item by item, the triple
    ChessDataset.__getitem__ (train.py:179-184) yields over the pickled lists -- bit for bit (planes, pi, z with its sign).
    Resume knows the compact files; a truncated tail (killed writer) is ignored."""
    import pickle
    from betaone_amd import records as R
    from betaone_amd import selfplay_main as M

    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 6
    config.DATA_DIR = str(tmp_path / "data")
    model = FakeNet(scale=2.0, salt=4)
    done = M.run_iteration(model, 3, n_games=5, n_slots=2, log=lambda s: None, records="both")
    assert sorted(done) == [0, 1, 2, 3, 4]
    path = R.compact_path(config.DATA_DIR, 3, 0)
    size = os.path.getsize(path)
    plies = sum(done.values())
    assert size < 400 * plies  # ~100-250 B per ply against 49.4 KB per ply in the pickles
    games = R.load_games(path)
    assert sorted(g["game_id"] for g in games) == [0, 1, 2, 3, 4]
    ds = R.CompactDataset([path], device="cpu", cache_games=2)
    assert len(ds) == plies
    k = 0
    for g in games:  # the dataset walks the games in file order
        dense = pickle.load(open(tmp_path / "data" / "iter_3" / f"game_{g['game_id']}.pkl", "rb"))
        assert len(dense) == g["n_plies"]
        for state, policy, value in dense:  # ChessDataset.__getitem__ of the reference over this list (train.py:179-184)
            want = (state, torch.from_numpy(policy).float(), torch.tensor([value], dtype=torch.float32))
            got = ds[k]
            assert got[0].dtype == torch.float32 and torch.equal(got[0], want[0])
            assert torch.equal(got[1], want[1]) and got[1].dtype == torch.float32
            assert torch.equal(got[2], want[2]) and np.signbit(got[2].numpy()[0]) == np.signbit(want[2].numpy()[0])
            k += 1
    assert k == len(ds) and torch.equal(ds[-1][0], ds[len(ds) - 1][0])
    with pytest.raises(IndexError):
        ds[len(ds)]
    # resume: compact-only mode skips what is in the file; a half-written game at the end is not counted
    assert M.pending_game_ids(config.DATA_DIR, 3, 7, records="compact") == [5, 6]
    with open(path, "ab") as fh:
        fh.write(open(path, "rb").read()[:100])
    assert len(R.load_games(path)) == 5 and R.game_ids_on_disk(config.DATA_DIR, 3) == {0, 1, 2, 3, 4}
    config.DATA_DIR = str(tmp_path / "data_c")
    only = M.run_iteration(model, 3, n_games=3, n_slots=3, log=lambda s: None, records="compact")
    assert sorted(only) == [0, 1, 2] and not list((tmp_path / "data_c" / "iter_3").glob("*.pkl"))
    a = {g["game_id"]: g for g in R.load_games(R.compact_path(config.DATA_DIR, 3, 0))}
    b = {g["game_id"]: g for g in games}
    for j in (0, 1, 2):  # the same games, whichever record form was asked for
        assert a[j]["moves"].tolist() == b[j]["moves"].tolist() and bytes(a[j]["positions"]) == bytes(b[j]["positions"])


def test_compact_file_survives_a_writer_killed_inside_a_write_and_both_mode_resumes_the_missing_form(tmp_path):
    """(i) a writer killed inside its write leaves a partial record at the tail; the next append must cut it off
    first, or every game appended behind it is read as the rest of that record and lost (N games, file chopped mid-record, M
    games appended -> N - 1 + M games back).  (ii) --records both: a game is finished only when BOTH forms are on disk; a run
    killed between the compact append and the pickle plays the game again and writes only the pickle (no duplicate record)."""
    from betaone_amd import records as R
    from betaone_amd import selfplay_main as M

    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 6
    config.DATA_DIR = str(tmp_path / "data")
    model = FakeNet(scale=2.0, salt=4)
    done = M.run_iteration(model, 3, n_games=4, n_slots=2, log=lambda s: None, records="both")
    assert sorted(done) == [0, 1, 2, 3]
    d = tmp_path / "data" / "iter_3"
    assert sorted(p.name for p in d.glob("*.pkl")) == [f"game_{j}.pkl" for j in range(4)] and not list(d.glob("*.tmp"))
    path = R.compact_path(config.DATA_DIR, 3, 0)
    blob = open(path, "rb").read()
    idx = R.scan_games(blob)
    assert len(idx) == 4
    order = [g[0] for g in idx]
    # (i) chop the file in the middle of its last record, then append two more games
    last_off, last_size = idx[-1][2], idx[-1][3]
    with open(path, "r+b") as fh:
        fh.truncate(last_off + last_size // 2)
    assert R.complete_prefix_bytes(path) == last_off and len(R.load_games(path)) == 3
    extra = R.unpack_games(blob)[:2]
    packed = [blob[g[2]:g[2] + g[3]] for g in idx[:2]]
    R.save_games(path, packed)
    got = R.load_games(path)
    assert [g["game_id"] for g in got] == order[:3] + order[:2]                 # N - 1 + M, nothing hidden behind the partial record
    assert os.path.getsize(path) == last_off + sum(len(b) for b in packed)
    assert got[-1]["moves"].tolist() == extra[1]["moves"].tolist()
    # (ii) the chopped game (compact form lost, pickle present) and a game whose pickle is lost (compact present) are both pending
    with open(path, "wb") as fh:
        fh.write(blob[:last_off])
    lost_compact, lost_pickle = order[3], order[0]
    os.remove(d / f"game_{lost_pickle}.pkl")
    assert sorted(M.pending_game_ids(config.DATA_DIR, 3, 4, records="both")) == sorted([lost_compact, lost_pickle])
    assert M.pending_game_ids(config.DATA_DIR, 3, 4, records="compact") == [lost_compact]
    again = M.run_iteration(model, 3, n_games=4, n_slots=2, log=lambda s: None, records="both")
    assert sorted(again) == sorted([lost_compact, lost_pickle])
    ids = [g["game_id"] for g in R.load_games(path)]
    assert sorted(ids) == [0, 1, 2, 3] and len(ids) == 4                         # no duplicate record of the game that only lacked its pickle
    assert (d / f"game_{lost_pickle}.pkl").exists() and M.pending_game_ids(config.DATA_DIR, 3, 4, records="both") == []
    replayed = {g["game_id"]: g for g in R.load_games(path)}[lost_compact]
    assert replayed["moves"].tolist() == R.unpack_games(blob)[3]["moves"].tolist()  # per-game seeds: the replay is the same game


def test_gpu_replay_buffer_yields_chessdataset_triples_and_evicts_oldest_games(tmp_path):
    """Row f3, the ingest side: records.GpuReplayBuffer keeps finished games as compact records on the device and expands training
    batches there (csrc/bo_replay.h).  Record i of the buffer (oldest resident game first) must be, bit for bit, item i of
    ChessDataset over the reference's pickles of the same games (train.py:179-184: planes with END-of-game repetition counts, dense
    pi, z with its sign) -- here against CompactDataset, which the test above pins to those pickles.  A full buffer evicts whole
    games, oldest first; the loader has DataLoader's contract for train_network's loop (train.py:252)."""
    from betaone_amd import records as R
    from betaone_amd import selfplay_main as M

    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 9
    config.DATA_DIR = str(tmp_path / "data")
    model = FakeNet(scale=2.0, salt=4)
    fens = None
    done = M.run_iteration(model, 1, n_games=7, n_slots=3, log=lambda s: None, records="compact")
    path = R.compact_path(config.DATA_DIR, 1, 0)
    games = R.load_games(path)
    ds = R.CompactDataset([path], device="cpu", cache_games=8)
    assert len(ds) == sum(done.values()) == 7 * 9
    buf = R.GpuReplayBuffer(capacity_plies=200, device="cpu")
    assert buf.add(games) == 0 and len(buf) == len(ds) and buf.n_games == 7
    st, pi, z = buf.batch(np.arange(len(ds)))
    assert st.dtype == pi.dtype == z.dtype == torch.float32 and tuple(st.shape) == (63, 120, 8, 8) and tuple(pi.shape) == (63, 4672) and tuple(z.shape) == (63, 1)
    for i in range(len(ds)):
        s0, p0, z0 = ds[i]
        assert torch.equal(st[i], s0) and torch.equal(pi[i], p0) and torch.equal(z[i], z0), i
        assert np.signbit(z[i].numpy()[0]) == np.signbit(z0.numpy()[0])
    one = buf.batch([40, 3, 40])                       # any order, repeats allowed
    assert torch.equal(one[0][0], ds[40][0]) and torch.equal(one[0][1], ds[3][0]) and torch.equal(one[1][2], ds[40][1])
    with pytest.raises(Exception):
        buf.batch([len(ds)])
    # DataLoader contract: an epoch visits every record once, in a seeded random order; `steps` batches are drawn with replacement
    seen = []
    loader = buf.loader(batch_size=16, seed=3)
    assert len(loader) == 4
    for states, policies, values in loader:
        assert states.shape[1:] == (120, 8, 8) and policies.shape[1] == 4672 and values.shape[1] == 1 and states.shape[0] == policies.shape[0] == values.shape[0] <= 16
        seen.append(states.shape[0])
    assert sum(seen) == 63 and seen[:-1] == [16, 16, 16]
    assert [b[0].shape[0] for b in buf.loader(batch_size=8, steps=5, seed=1)] == [8] * 5
    # a small buffer: 25 plies + slack -> position slots for two 9-ply games (10 slots each) and a bit: whole games leave, oldest first
    small = R.GpuReplayBuffer(capacity_plies=25, device="cpu")
    lost = [small.add([g]) for g in games]
    assert lost == [0, 0, 9, 9, 9, 9, 9] and len(small) == 18 and small.n_games == 2 and small.n_evicted == 45
    st2, pi2, z2 = small.batch(np.arange(18))
    off = 5 * 9                                        # the two newest games = the last 18 items of the dataset
    for i in range(18):
        assert torch.equal(st2[i], ds[off + i][0]) and torch.equal(pi2[i], ds[off + i][1]) and torch.equal(z2[i], ds[off + i][2]), i
    with pytest.raises(Exception, match="longer than the buffer"):
        R.GpuReplayBuffer(capacity_plies=4, device="cpu").add(games[:1])
    buf.close(); small.close()


def test_weights_are_swapped_inside_a_living_process(tmp_path):
    """Row f4: main.py:147-148 hands new weights to its workers through best_model.pth.  ModelFileWatcher notices the changed file,
    run_self_play_games swaps the evaluate stage between two plies (Rollout.swap_model): plies before the swap are those of the
    old weights, plies after it those of the new ones -- checked against two runs that never swap."""
    import time as _time
    from betaone_amd import selfplay_main as M

    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 8

    class Net(torch.nn.Module):  # a FakeNet with a state_dict: the salt is its one "weight"
        def __init__(self, salt=1):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([float(salt)]), requires_grad=False)

        def forward(self, x):
            return FakeNet(scale=3.0, salt=int(self.w.item()))(x)

    path = tmp_path / "best_model.pth"
    torch.save(Net(1).state_dict(), path)
    old = Net(1)
    old.load_state_dict(torch.load(path))
    calls = [0]
    watcher = M.ModelFileWatcher(str(path), Net, "cpu", every=1)

    def reload_model():
        calls[0] += 1
        if calls[0] == 4:  # the training side writes new weights while the games are at their 4th ply
            _time.sleep(0.01)
            torch.save(Net(2).state_dict(), path)
        return watcher.poll()

    mixed = self_play.run_self_play_games(old, [0, 1, 2], seeds=[5, 6, 7], n_slots=3, reload_model=reload_model)
    assert watcher.n_reloads == 1
    only_old = self_play.run_self_play_games(Net(1), [0, 1, 2], seeds=[5, 6, 7], n_slots=3)
    differs = 0
    for j in (0, 1, 2):
        assert len(mixed[j]) == len(only_old[j]) == 8
        for k in range(3):  # plies searched before the swap: the old weights' (state, pi)
            assert torch.equal(mixed[j][k][0], only_old[j][k][0]) and np.array_equal(mixed[j][k][1], only_old[j][k][1])
        differs += any(not np.array_equal(mixed[j][k][1], only_old[j][k][1]) or not torch.equal(mixed[j][k][0], only_old[j][k][0])
                       for k in range(3, 8))
    assert differs >= 1  # ... and the new weights took over afterwards
    # an unchanged file is not reloaded; a file that cannot be loaded yet is retried
    w2 = M.ModelFileWatcher(str(path), Net, "cpu", every=1)
    assert w2.poll() is None
    _time.sleep(0.01)
    path.write_bytes(b"half a file")
    assert w2.poll() is None and w2.n_reloads == 0
    torch.save(Net(3).state_dict(), path)
    assert int(w2.poll().w.item()) == 3 and w2.n_reloads == 1


@pytest.mark.parametrize("cohorts", [2, 3])
def test_cohorts_of_games_play_exactly_the_games_one_rollout_plays(cohorts):
    """rollout.CohortRollout (config.COHORTS): the resident games as K phase-shifted cohorts, each a Rollout with its own engine,
    driven through ply_begin / ply_end with the plies software-pipelined.  Games are independent (main.py:160-175 runs them in
    separate processes), so every game must come out exactly as from one Rollout: states, pi, z -- with refills, games of uneven
    length (a position that is over at once, a game from a FEN) and more games than slots."""
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 30, 16, 7
    model = FakeNet(scale=3.0, salt=9)
    ids = list(range(11))
    seeds = [40 + i for i in ids]
    mate = "k6R/8/1K6/8/8/8/8/8 b - - 1 1"
    fens = [None, None, mate, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"] + [None] * 7
    seen = []
    config.COHORTS = 1
    one = self_play.run_self_play_games(model, ids, seeds=seeds, n_slots=6, start_fens=fens)
    config.COHORTS, config.COHORT_MIN_SLOTS = cohorts, 1
    try:
        many = self_play.run_self_play_games(model, ids, seeds=seeds, n_slots=6, start_fens=fens, on_game=lambda fin: seen.append((fin.game_id, fin.slot)))
    finally:
        config.COHORTS = 1
        del config.COHORT_MIN_SLOTS
    assert sorted(many) == sorted(one) == ids and many[2] == one[2] == []
    assert sorted(g for g, _ in seen) == ids and {s for _, s in seen} <= set(range(6)) and len({s for _, s in seen}) > 6 // cohorts  # global slot numbers
    for g in ids:
        assert len(many[g]) == len(one[g])
        for (s1, p1, z1), (s2, p2, z2) in zip(many[g], one[g]):
            assert torch.equal(s1, s2) and np.array_equal(p1, p2) and z1 == z2 and np.signbit(z1) == np.signbit(z2)


def test_cohort_count_is_the_largest_share_of_config_cohorts_that_keeps_64_slots_per_cohort():
    """dropin.self_play._cohorts: config.COHORTS is an upper limit, halved until it divides the slot count and leaves every cohort
    COHORT_MIN_SLOTS (64) slots; one Rollout for the Python-RNG mode and for the fast search."""
    saved = config.COHORTS
    try:
        config.COHORTS = 4
        assert [self_play._cohorts(n, "native") for n in (256, 2048, 192, 128, 130, 64, 6)] == [4, 4, 2, 2, 2, 1, 1]
        assert self_play._cohorts(256, "python") == 1
        config.COHORTS = 1
        assert self_play._cohorts(256, "native") == 1
    finally:
        config.COHORTS = saved


def test_prefetched_result_block_and_fetched_one_give_the_same_games(monkeypatch):
    """Rollout._prefetch_result (bo_search_result_prefetch + bo_selfplay_turn flag 8): the result block enqueued behind the searches by
    ply_begin, or fetched by the turn itself -- the same games either way, also when a search needs more evaluations than were enqueued
    (expected_evals lowered: the turn then finds searches still running, steps them and prefetches again)."""
    from betaone_amd import rollout as R
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 40, 8, 6
    model = FakeNet(scale=2.0, salt=3)
    ids, seeds = list(range(5)), [70 + i for i in range(5)]
    runs = []
    for prefetch, short in ((True, False), (False, False), (True, True)):
        monkeypatch.setattr(R.Rollout, "PREFETCH_RESULT", prefetch)
        if short:  # every ply: one evaluation fewer enqueued than the searches need
            orig = R.Rollout.__init__

            def init(self, *a, **k):
                orig(self, *a, **k)
                self.expected_evals -= 1
            monkeypatch.setattr(R.Rollout, "__init__", init)
        config.COHORTS, config.COHORT_MIN_SLOTS = 2, 1
        try:
            runs.append(self_play.run_self_play_games(model, ids, seeds=seeds, n_slots=4))
        finally:
            config.COHORTS = 1
            del config.COHORT_MIN_SLOTS
    for other in runs[1:]:
        for g in ids:
            assert len(other[g]) == len(runs[0][g]) > 0
            for (s1, p1, z1), (s2, p2, z2) in zip(other[g], runs[0][g]):
                assert torch.equal(s1, s2) and np.array_equal(p1, p2) and z1 == z2


def test_one_overlong_game_does_not_end_the_others():
    """config.ENGINE_MAX_PLIES smaller than the games: a slot whose position stack is full stops THAT game like the
    reference's move limit (self_play.py:186: records of the moves played are kept) and every other game -- running or
    still queued -- is played to its own end (the reference plays one game per call, so games cannot affect each other)."""
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 24, 8, 6
    model = FakeNet(scale=0.0, salt=13)
    full = self_play.run_self_play_games(model, [0, 1, 2, 3, 4], seeds=[20, 21, 22, 23, 24], n_slots=2)
    assert all(len(full[g]) == 6 for g in range(5))
    # game 1 starts from a 3-move prefix-free FEN deep into a game; capacity 5 positions = at most 3 moves per slot
    config.ENGINE_MAX_PLIES = 5
    try:
        cut = self_play.run_self_play_games(model, [0, 1, 2, 3, 4], seeds=[20, 21, 22, 23, 24], n_slots=2)
    finally:
        config.ENGINE_MAX_PLIES = None
    assert sorted(cut) == [0, 1, 2, 3, 4]
    for g in range(5):  # every game reported, each with the records of the moves that fitted, equal to the full run's prefix
        assert cut[g] is not None and 1 <= len(cut[g]) < 6
        for (s1, p1, _z1), (s2, p2, _z2) in zip(cut[g], full[g]):
            assert np.array_equal(p1, p2)
            assert torch.equal(s1[98:112], s2[98:112])   # the current-position block; repetition planes of older blocks use the end-of-game tracker


def test_overlong_games_keep_the_same_records_under_cohorts():
    """a game retired for a full position stack (ENGINE_MAX_PLIES) is found while its cohort's NEXT ply is already
    enqueued; CohortRollout.retire ends that ply without a turn for the leaving game (ply_end(drop=...)), so the game keeps exactly the
    records the single Rollout gives it -- same number of examples, same pi, same outcome signs -- and every other game is untouched."""
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 24, 8, 7
    model = FakeNet(scale=1.0, salt=29)
    ids, seeds = list(range(7)), [40 + i for i in range(7)]
    config.ENGINE_MAX_PLIES = 6
    runs = []
    try:
        for cohorts in (1, 2):
            config.COHORTS, config.COHORT_MIN_SLOTS = cohorts, 1
            runs.append(self_play.run_self_play_games(model, ids, seeds=seeds, n_slots=4))
    finally:
        config.ENGINE_MAX_PLIES = None
        config.COHORTS = 1
        del config.COHORT_MIN_SLOTS
    one, two = runs
    assert sorted(one) == sorted(two) == ids
    for g in ids:
        assert one[g] is not None and two[g] is not None and 1 <= len(one[g]) < 7
        assert len(one[g]) == len(two[g]), g
        for (s1, p1, z1), (s2, p2, z2) in zip(one[g], two[g]):
            assert torch.equal(s1, s2) and np.array_equal(p1, p2) and z1 == z2 and np.signbit(z1) == np.signbit(z2)


def test_game_from_a_finished_position_yields_no_examples():
    """A start position that is already over: the reference's loop body never runs and the game returns zero examples
    (self_play.py:101,200-216); the batched native-RNG path must not attach a stale pi of the slot's previous occupant."""
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 24, 8, 3
    model = FakeNet(scale=0.0, salt=17)
    mate = "k6R/8/1K6/8/8/8/8/8 b - - 1 1"
    out = self_play.run_self_play_games(model, [0, 1, 2], seeds=[1, 2, 3], n_slots=1, start_fens=[None, mate, None])
    assert len(out[0]) == 3 and len(out[2]) == 3
    assert out[1] == []


@pytest.mark.product_backend
def test_device_string_without_index_resolves_to_the_current_device(monkeypatch):
    """Every rank of a torchrun job passes config.DEVICE == 'cuda' after torch.cuda.set_device(LOCAL_RANK): the engine,
    its NN rows and the model must all land on THAT GPU (not on GPU 0)."""
    from betaone_amd import engine as E

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert E.runtime_device("cuda") == torch.device("cuda", 3)
    assert E.runtime_device(torch.device("cuda")) == torch.device("cuda", 3)
    assert E.runtime_device("cuda:1") == torch.device("cuda", 1)
    with pytest.raises(E.EngineError):
        E.runtime_device("cpu")


@pytest.mark.parametrize("stop_after_evals", [0, 1, 3, 6])
def test_interrupted_search_equals_a_complete_search_of_the_simulations_done(stop_after_evals):
    """Row f2: a search interrupted between two steps returns exactly what the reference returns for NUM_SIMULATIONS = the
    simulations completed so far (tail batch flushed as in mcts.py:256-257, the outstanding evaluation dropped)."""
    config.NUM_SIMULATIONS = 800
    fen, moves = chess.STARTING_FEN, "e2e4 e7e5 g1f3 b8c6 f1b5".split()
    board, history, tracker = context(fen, moves)

    class StoppingNet(FakeNet):
        def __call__(self, x):
            if len(self.calls) >= stop_after_evals:
                mcts.request_stop()
            return super().__call__(x)

    mcts.stop_event.clear()
    try:
        np.random.seed(3)
        best, pi = mcts.run_mcts(board, StoppingNet(scale=0.0, salt=8), history, tracker)
    finally:
        mcts.stop_event.clear()
    info = dict(mcts.last_search)
    assert info["stopped"] and info["simulations"] < 800
    assert info["simulations"] == stop_after_evals * 96    # whole batches: 96 rows share one evaluation (E1)
    ob, oh, ot = oracle_context(fen, moves)
    r = O.run_mcts(ob, oh, ot, uniform_eval(8), np.random.RandomState(3), O.default_config(num_simulations=info["simulations"]))
    assert best.uci() == O.move_to_uci(r["best"])
    assert np.array_equal(pi.view(np.uint32), r["pi"].view(np.uint32))
    # and an uninterrupted call afterwards is a full search again
    np.random.seed(3)
    mcts.run_mcts(board, FakeNet(scale=0.0, salt=8), history, tracker)
    assert mcts.last_search == {"simulations": 800, "stopped": False}
