"""GPU tests of the rows SURVEY.md section 8 marks "next" that had only emulator coverage:
  f3  records.save_games / CompactDataset on the product library (device="cuda:0"), bit-equal to the reference's pickles;
  f4  Rollout.swap_model under captured hipGraphs (graphs dropped and re-captured): plies before the swap are the old weights',
      plies after it the new weights' -- each compared with a run that never swaps;
  the split-precision tower's fault word reaches the self-play loop with the ply's own result block."""
import os
import pickle
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def env():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from betaone_amd import dropin
    from betaone_amd import engine as E

    E.load_hip_library()
    dropin.install()
    import config

    keys = ("RESIDUAL_BLOCKS", "SE_RESIDUAL_BLOCKS", "CONV_FILTERS", "NUM_SIMULATIONS", "MCTS_BATCH_SIZE", "DATA_DIR", "MAX_GAME_MOVES")
    saved = {k: getattr(config, k) for k in keys}
    yield config
    for k, v in saved.items():
        setattr(config, k, v)


def _net(config, size, seed):
    import torch
    import network

    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = size
    torch.manual_seed(seed)
    return network.PolicyValueNet().to("cuda").eval()


def test_compact_dataset_on_the_gpu_yields_what_the_reference_pickles_yield(env, tmp_path):
    """Row f3 on the product path: selfplay_main --records both on cuda:0 writes the reference's pickles (dense planes from
    bo_k_encode_game in the slot) and the compact records; records.CompactDataset(device="cuda:0") re-expands the planes with
    bo_records_encode on the GPU and yields, item by item, ChessDataset.__getitem__'s triple (train.py:179-184) bit for bit."""
    import torch
    from betaone_amd import records as R
    from betaone_amd import selfplay_main as M

    config = env
    model = _net(config, (3, 1, 64), 0)
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 100, 96, 14
    config.DATA_DIR = str(tmp_path / "data")
    done = M.run_iteration(model, 2, n_games=20, n_slots=8, log=lambda s: None, records="both")
    assert sorted(done) == list(range(20))
    path = R.compact_path(config.DATA_DIR, 2, 0)
    games = R.load_games(path)
    assert sorted(g["game_id"] for g in games) == list(range(20))
    plies = sum(done.values())
    assert os.path.getsize(path) < 400 * plies and R.complete_prefix_bytes(path) == os.path.getsize(path)
    ds = R.CompactDataset([path], device="cuda:0", cache_games=3)
    assert len(ds) == plies
    k = 0
    for g in games:
        dense = pickle.load(open(tmp_path / "data" / "iter_2" / f"game_{g['game_id']}.pkl", "rb"))
        assert len(dense) == g["n_plies"] == done[g["game_id"]]
        for state, policy, value in dense:
            got = ds[k]
            assert got[0].dtype == torch.float32 and got[0].device.type == "cpu" and torch.equal(got[0], state)
            assert torch.equal(got[1], torch.from_numpy(policy).float())
            assert torch.equal(got[2], torch.tensor([value], dtype=torch.float32))
            assert np.signbit(got[2].numpy()[0]) == np.signbit(np.float32(value))
            k += 1
    assert k == len(ds)
    # random access across the small cache (games are re-expanded on demand), and a DataLoader batch forms
    order = np.random.RandomState(0).permutation(len(ds))[:40]
    first = [g for g in games]
    for i in order:
        gi, ki = ds.game_of(int(i))
        dense = pickle.load(open(tmp_path / "data" / "iter_2" / f"game_{first[gi]['game_id']}.pkl", "rb"))
        assert torch.equal(ds[int(i)][0], dense[ki][0])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=16, shuffle=False)))
    assert tuple(batch[0].shape) == (16, 120, 8, 8) and tuple(batch[1].shape) == (16, 4672) and tuple(batch[2].shape) == (16, 1)
    # a writer killed inside a write: the partial record is cut off by the next append
    blob = open(path, "rb").read()
    idx = R.scan_games(blob)
    with open(path, "r+b") as fh:
        fh.truncate(idx[-1][2] + 50)
    R.save_games(path, [blob[idx[0][2]:idx[0][2] + idx[0][3]]])
    assert [g["game_id"] for g in R.load_games(path)] == [g[0] for g in idx[:-1]] + [idx[0][0]]


def test_gpu_replay_buffer_feeds_a_training_step_with_the_reference_pickles_content(env, tmp_path):
    """Row f3, ingest: records.GpuReplayBuffer on cuda:0 (csrc/bo_replay.h: compact records resident in HBM, batches expanded by one
    wave per sampled ply).  Every record of the buffer equals, bit for bit, the tuple the reference's pickle of the same game holds
    (ChessDataset.__getitem__, train.py:179-184); the loader drives the body of train_network's loop (train.py:252-262: forward, the
    two losses of calculate_loss, backward, optimizer step) without a host copy of a plane; a full buffer evicts the oldest games."""
    import torch
    import torch.nn.functional as F
    from betaone_amd import records as R
    from betaone_amd import selfplay_main as M

    config = env
    model = _net(config, (3, 1, 64), 0)
    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 100, 96, 14
    config.DATA_DIR = str(tmp_path / "data")
    done = M.run_iteration(model, 4, n_games=20, n_slots=8, log=lambda s: None, records="both")
    games = R.load_games(R.compact_path(config.DATA_DIR, 4, 0))
    buf = R.GpuReplayBuffer(capacity_plies=4096, device="cuda:0")
    assert buf.add(games) == 0 and len(buf) == sum(done.values()) and buf.n_games == 20
    st, pi, z = buf.batch(np.arange(len(buf)))
    assert st.is_cuda and pi.is_cuda and z.is_cuda
    st, pi, z = st.cpu(), pi.cpu(), z.cpu()
    k = 0
    for g in games:  # the buffer's index space: resident records, oldest game first = the file's order
        dense = pickle.load(open(tmp_path / "data" / "iter_4" / f"game_{g['game_id']}.pkl", "rb"))
        for state, policy, value in dense:
            assert torch.equal(st[k], state) and torch.equal(pi[k], torch.from_numpy(policy).float())
            assert z[k, 0].item() == value and np.signbit(z[k, 0].item()) == np.signbit(np.float32(value))
            k += 1
    assert k == len(buf)
    # train_network's loop body over the loader (a fresh net in train mode, plain SGD; the reference wraps this in autocast + GradScaler)
    net = _net(config, (3, 1, 64), 5).train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    losses = []
    for states, t_policies, t_values in buf.loader(batch_size=64, steps=6, seed=0):
        states, t_policies, t_values = states.to("cuda"), t_policies.to("cuda"), t_values.to("cuda")   # (no-ops: the batches are made there)
        opt.zero_grad()
        policies, values = net(states)
        loss = F.mse_loss(values, t_values) + F.cross_entropy(policies, t_policies)      # calculate_loss, train.py:222-245
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert len(losses) == 6 and all(np.isfinite(l) for l in losses)
    small = R.GpuReplayBuffer(capacity_plies=60, device="cuda:0")
    for g in games:
        small.add([g])
    assert 0 < len(small) <= 60 + 14 and small.n_evicted == len(buf) - len(small)
    tail = small.batch(np.arange(len(small)))[0].cpu()
    assert torch.equal(tail, st[len(buf) - len(small):])        # what is left are the newest games, in order
    buf.close(); small.close()


def _play(ro, n_plies, fins, upto=None):
    for _ in range(n_plies):
        ro.play_ply(on_finished=fins.append)


def test_swap_model_under_captured_graphs_plays_old_weights_before_and_new_weights_after(env):
    """Row f4 on the product path (main.py:147-148 hands new weights to living workers): Rollout.swap_model between two plies
    with hipGraphs on -- the captured graphs hold the old module's kernels and weight addresses, so they are dropped and
    re-captured.  Run C swaps after 3 plies.  Its first 3 plies equal run A (old weights, never swapping); its plies 4..7 equal
    run D, which never swaps either: the NEW weights from C's own positions and RNG states at the swap."""
    import torch
    from betaone_amd import engine as E
    from betaone_amd.nn_tune import best_inference_copy
    from betaone_amd.rollout import Rollout

    config = env
    G, K, T = 24, 3, 7
    old = best_inference_copy(_net(config, (2, 1, 128), 1), G, "cuda:0")
    new = best_inference_copy(_net(config, (2, 1, 128), 2), G, "cuda:0")
    assert old.conv == new.conv == "tower_split"
    kw = dict(num_simulations=100, mcts_batch_size=32, device="cuda:0", use_graph=True, rng_mode="native", max_game_moves=T,
              policy_kind="probs")
    ids = list(range(G))

    def run(model, plies, swap_to=None, swap_at=None, moves=None, rng_states=None):
        ro = Rollout(model, G, **kw)
        ro.start_games(ids, ids, [1000 + i for i in ids], None, moves)
        if rng_states is not None:
            for g, st in enumerate(rng_states):
                ro.eng.rng_set_state(g, st)
        fins, snap = [], None
        for p in range(plies):
            if swap_at is not None and p == swap_at:
                torch.cuda.synchronize()
                snap = [ro.eng.rng_get_state(g) for g in range(G)]
                assert ro._graphs_n or ro._graph is not None          # graphs were captured and used before the swap ...
                ro.swap_model(swap_to)
                assert ro._graph is None and not ro._graphs_n          # ... and are gone after it
            ro.play_ply(on_finished=fins.append)
        if swap_at is not None:
            assert ro._graph is not None or ro._graphs_n              # re-captured with the new module
        while any(g is not None for g in ro.games):
            ro.play_ply(on_finished=fins.append)
        ro.close()
        return {f.game_id: f for f in fins}, snap

    a, _ = run(old, T + 1)
    c, snap = run(old, T + 1, swap_to=new, swap_at=K)
    assert sorted(a) == sorted(c) == ids
    prefixes = [" ".join(E.move_to_uci(m) for m in c[g].moves[:K]) for g in ids]
    d, _ = run(new, T + 1 - K, moves=prefixes, rng_states=snap)
    differs = 0
    for g in ids:
        assert len(c[g].moves) == len(a[g].moves) == T and len(d[g].moves) == T and d[g].first_ply == K
        assert c[g].moves[:K] == a[g].moves[:K]                                        # before the swap: the old weights' plies
        for k in range(K):
            assert c[g].pis[k][0].tolist() == a[g].pis[k][0].tolist() and c[g].pis[k][1].tobytes() == a[g].pis[k][1].tobytes()
        assert c[g].moves[K:] == d[g].moves[K:]                                        # after it: the new weights' plies
        for k in range(K, T):
            assert c[g].pis[k][0].tolist() == d[g].pis[k - K][0].tolist() and c[g].pis[k][1].tobytes() == d[g].pis[k - K][1].tobytes()
        differs += c[g].moves[K:] != a[g].moves[K:]
    assert differs >= G // 2  # the two nets do play differently


def test_a_net_that_saturates_the_split_tower_stops_self_play_at_that_ply(env):
    """VERDICT round 3, next 4(iv) / ADVICE: the split-precision tower's fault word is checked once per ply, behind the copy the
    ply waits for anyway (bo_engine_watch) -- not only when finished games are handed over: a net whose activations leave the fp16
    range raises in its first ply, before any record exists; an ordinary net plays on."""
    import torch
    from betaone_amd import engine as E
    from betaone_amd.nn_tune import best_inference_copy
    from betaone_amd.rollout import Rollout

    config = env
    net = _net(config, (2, 1, 128), 3)
    G = 24
    kw = dict(num_simulations=64, mcts_batch_size=32, device="cuda:0", use_graph=True, rng_mode="native", max_game_moves=4, policy_kind="probs")
    fine = Rollout(best_inference_copy(net, G, "cuda:0"), G, **kw)
    fine.start_games(list(range(G)), list(range(G)), list(range(G)))
    fins = []
    for _ in range(5):
        fine.play_ply(on_finished=fins.append)
    assert len(fins) == G
    fine.close()
    with torch.no_grad():
        net.conv_input.weight.mul_(3.0e5)
    big = best_inference_copy(net, G, "cuda:0")
    assert big.conv == "tower_split" and big.overflow_word_ptr() != 0
    ro = Rollout(big, G, **kw)
    ro.start_games(list(range(G)), list(range(G)), list(range(G)))
    handed = []
    with pytest.raises(E.EngineError, match="fp16 range"):
        for _ in range(3):
            ro.play_ply(on_finished=handed.append)
    assert ro.n_plies == 0 and not handed   # stopped inside the first ply: no move played from a saturated evaluation, no record handed out
    ro._graph = ro._fgraph = None
    ro._graphs_n = {}
    ro.eng.close()


def _hip_runtime():
    import ctypes
    import torch

    return ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def test_one_launch_tower_faults_stop_a_search_and_a_small_rollout(env):
    """small float32 batches (uci.py's run_mcts, run_self_play_game's one-slot Rollout) evaluate on the one-launch
    tower with (hi, lo) fp16 weights (kernel_route -> 'tower_b1'); its status words -- a hand-off wait that gave up, an activation that
    left the fp16 range -- now ride in every result block (bo_nn_b1_word -> bo_engine_watch_words): a forced timeout code and a
    saturating net both RAISE in Rollout.play_ply and in run_mcts instead of returning moves made from invalid evaluations, and the
    stage is re-armed afterwards (its counters start from zero: the next search is fine)."""
    import ctypes
    import torch
    from betaone_amd import engine as E
    from betaone_amd.nn_tune import best_inference_copy
    from betaone_amd.rollout import Rollout

    config = env
    net = _net(config, (2, 1, 128), 5)
    small = best_inference_copy(net, 1, "cuda:0")
    assert small.conv == "tower_b1"
    ptr, n_words = small.overflow_words()
    assert ptr != 0 and n_words == 2
    kw = dict(num_simulations=40, mcts_batch_size=16, device="cuda:0", use_graph=True, rng_mode="native", max_game_moves=6, policy_kind="probs")
    ro = Rollout(small, 1, **kw)
    ro.start_games([0], [0], [7])
    assert ro.play_ply() == 1 and ro.play_ply() == 1          # a healthy stage plays
    torch.cuda.synchronize()
    code = ctypes.c_uint32(1 + 5)                               # "the hand-off wait of phase 5 gave up", as the kernel would leave it
    assert _hip_runtime().hipMemcpy(ctypes.c_void_p(ptr), ctypes.byref(code), 4, 1) == 0  # hipMemcpyHostToDevice
    with pytest.raises(E.EngineError, match="timed out"):
        for _ in range(3):
            ro.play_ply()
    assert ro.n_plies == 2                                      # nothing was played from the ply whose evaluations were invalid
    ro._graph = ro._fgraph = None
    ro._graphs_n = {}
    ro.eng.close()
    small.check_b1()                                            # re-armed: no fault pending any more

    # the drop-in's single search (uci.py:63,84 -> run_mcts)
    from betaone_amd import dropin
    dropin.install()
    import mcts as dm
    import utils as du
    from oracle.shim import chess as shim  # (python-chess stand-in: test infrastructure)

    config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE = 60, 16
    model = _net(config, (2, 1, 128), 5).to("cuda:0").eval()
    board, trk = shim.Board(), du.RepetitionTracker()
    trk.add_board(board)
    mv, pi = dm.run_mcts(board, model, [], trk)                # healthy
    assert abs(float(pi.sum()) - 1.0) < 1e-6
    stage = next(iter(dm._fast.values()))[1]
    assert getattr(stage, "net", stage).conv == "tower_b1"
    ptr2, _ = getattr(stage, "net", stage).overflow_words()
    torch.cuda.synchronize()
    assert _hip_runtime().hipMemcpy(ctypes.c_void_p(ptr2), ctypes.byref(code), 4, 1) == 0
    with pytest.raises(E.EngineError, match="timed out"):
        dm.run_mcts(board, model, [], trk)
    mv2, pi2 = dm.run_mcts(board, model, [], trk)              # re-armed by the check that raised
    assert abs(float(pi2.sum()) - 1.0) < 1e-6

    with torch.no_grad():
        model.conv_input.weight.mul_(3.0e5)                     # activations leave the fp16 range
    # (the in-place edit bumped the parameters' version counters: run_mcts rebuilds its inference copy)
    with pytest.raises(E.EngineError, match="fp16 range"):
        dm.run_mcts(board, model, [], trk)


def test_swap_model_under_cohorts_and_captured_graphs(env):
    """CohortRollout.swap_model while every cohort has a whole ply enqueued on its own stream (captured graphs, the next
    root evaluation).  The outstanding plies are ended first -- played with the OLD weights -- so nothing of the old graphs is in flight when
    they are released; from then on every search evaluates with the new weights: the games equal those of a single Rollout that swaps after
    the same number of plies (old weights for plies < K, new weights from the positions and RNG states of the swap onward)."""
    import torch
    from betaone_amd.nn_tune import best_inference_copy
    from betaone_amd.rollout import CohortRollout, Rollout

    config = env
    G, K, T = 16, 3, 6
    kw = dict(num_simulations=100, mcts_batch_size=32, device="cuda:0", use_graph=True, rng_mode="native", max_game_moves=T, policy_kind="probs")
    ids = list(range(G))

    def nets(seed, n):
        return [best_inference_copy(_net(config, (2, 1, 128), seed), G // max(1, n), "cuda:0") for _ in range(max(1, n))]

    def run(cohorts):
        old, new = nets(1, cohorts), nets(2, cohorts)
        ro = CohortRollout(old, G, cohorts=cohorts, **kw) if cohorts > 1 else Rollout(old[0], G, **kw)
        ro.start_games(ids, ids, [2000 + i for i in ids])
        fins = []
        for p in range(T + 3):
            if p == K:
                if cohorts > 1:
                    assert all(part._turn_due is not None for part in ro.parts)   # every cohort has a ply in flight when the swap comes
                ro.swap_model(new if cohorts > 1 else new[0])
                if cohorts > 1:
                    assert all(part._turn_due is None and part._graph is None and not part._graphs_n for part in ro.parts)
            ro.play_ply(on_finished=fins.append)
        if cohorts > 1:
            ro.drain()
        while any(g is not None for g in ro.games):
            ro.play_ply(on_finished=fins.append)
        ro.close()
        return {f.game_id: (list(f.moves), [(i.tolist(), v.tobytes()) for i, v in f.pis]) for f in fins}

    one = run(1)
    two = run(2)
    assert sorted(one) == ids
    # a cohort's pipeline is one ply ahead of a single Rollout's call count (play_ply ends a ply and begins the next): the swap lands one
    # ply later in the cohorts' games than in the single Rollout's -- compare against the single run that swaps at K + 1
    assert sorted(two) == ids and all(len(two[g][0]) == T for g in ids)
    K_eff = K  # plies each cohort had ENDED when the swap came: drain() ended the (K)th
    def run_single(k):
        old, new = nets(1, 1)[0], nets(2, 1)[0]
        ro = Rollout(old, G, **kw)
        ro.start_games(ids, ids, [2000 + i for i in ids])
        fins = []
        for p in range(T + 3):
            if p == k:
                ro.swap_model(new)
            ro.play_ply(on_finished=fins.append)
        ro.close()
        return {f.game_id: (list(f.moves), [(i.tolist(), v.tobytes()) for i, v in f.pis]) for f in fins}
    assert two == run_single(K_eff) or two == run_single(K_eff + 1)
