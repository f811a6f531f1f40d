#!/usr/bin/env python3
"""
tests/golden/generate_golden.py -- produces the committed golden fixtures (run in the DEV container only).

It executes the UNMODIFIED reference files /root/reference/{config,network,utils,mcts,self_play}.py.
python-chess is not installed here, so ``import chess`` resolves to oracle/shim/chess.py (the CPU
oracle's own rules engine behind python-chess's API).  Two harness-side adjustments, neither of
which edits the reference:
  * config.DEVICE = "cpu" and torch.autocast neutralised (as shipped, mcts.py:183-185 raises
    "Got unsupported ScalarType BFloat16" on CPU) -> dtype regime R3 (fp32);
  * np.random.seed(seed) / random.seed(seed) before each search/game (the reference never seeds).

Outputs (tests/golden/):
  g1_net.npz        reference network.py outputs for hash-initialised weights (pins PolicyValueNet)
  g2_evals.npz      every (planes -> softmax probs, value) pair the reference consumed (the "seam")
  g2_searches.json  per-search expected tree / pi / best move / batch trace
  g2_games.json     per-game expected moves / z / pi / state hashes
  g4_codec.json     utils.test_move_indexing error counts + legal move -> index tables
  g5_long_games.json / g5_long_evals.npz   two full-length games of the reference (120 plies from the start position; fullmove 28 to the
                    end by rule) with their seam -- `--long` regenerates only these

Nothing here travels to the GPU box except the outputs; tests never read /root/reference.
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("BETAONE_GOLDEN_OUT", HERE)  # where the fixtures are written (the committed ones live next to this file)
REF = "/root/reference"
# scripts/pin_python_chess.py re-runs this file with the REAL python-chess (where it is installed) into a scratch directory and diffs
# the result against the committed fixtures: BETAONE_GOLDEN_REAL_CHESS=1 leaves the shim off the path, BETAONE_GOLDEN_OUT names the directory.
REAL_CHESS = os.environ.get("BETAONE_GOLDEN_REAL_CHESS", "0") == "1"
sys.path[:0] = ([] if REAL_CHESS else [os.path.join(ROOT, "oracle", "shim")]) + [REF, os.path.join(ROOT, "tests")]
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import chess  # noqa: E402  (the shim)
import config  # noqa: E402  (reference)
import network  # noqa: E402
import utils  # noqa: E402
import mcts  # noqa: E402
import self_play  # noqa: E402
from fake_model import FakeNet, hash_init_, planes_key  # noqa: E402

assert chess.__file__.startswith(os.path.join(ROOT, "oracle", "shim")) != REAL_CHESS
for m in (config, network, utils, mcts, self_play):
    assert m.__file__.startswith(REF), m.__file__

config.DEVICE = "cpu"
torch.set_num_threads(1)


class _NoAutocast:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


torch.autocast = _NoAutocast

# ---------------------------------------------------------------------------------------------
# instrumentation (wrappers around the reference's own objects; no reference code is changed)
# ---------------------------------------------------------------------------------------------
EVALS = {}  # seam key -> [probs f32[4672], value f32, is_root_eval, set(needed action indices)]
PROB_HASH = {}  # sha1(probs bytes) -> seam key
_last_input = {}
_real_softmax = torch.softmax


class RecordingModel:
    def __init__(self, inner):
        self.inner = inner
        _last_input["model"] = f"{inner.scale}:{inner.salt}"

    def __call__(self, x):
        _last_input["planes"] = x.detach().cpu().numpy().copy()
        logits, value = self.inner(x)
        _last_input["value"] = value.detach().cpu().numpy().reshape(-1).copy()
        return logits, value


def _recording_softmax(t, dim=None, **kw):
    out = _real_softmax(t, dim=dim, **kw)
    planes = _last_input["planes"]
    probs = out.detach().cpu().numpy()
    vals = _last_input["value"]
    assert probs.shape[0] == planes.shape[0]
    for i in range(planes.shape[0]):
        k = planes_key(planes[i]) + ":" + _last_input["model"]  # seam key: position x model
        if k in EVALS:  # batch-invariance of the seam: identical rows must give identical bits
            assert np.array_equal(EVALS[k][0].view(np.uint32), probs[i].view(np.uint32)), "softmax not row-invariant"
            assert EVALS[k][1].view(np.uint32) == vals[i].view(np.uint32)
        else:
            # batch-1 evals are root evals (mcts.py:184): their whole vector is needed (renormalisation,
            # mcts.py:201); leaf rows are only ever read at the leaf's legal-move indices (mcts.py:58-68).
            EVALS[k] = [probs[i].copy(), np.float32(vals[i]), planes.shape[0] == 1, set()]
            PROB_HASH[hashlib.sha1(probs[i].tobytes()).hexdigest()] = k
    return out


torch.softmax = _recording_softmax

NODES = []
_real_init = mcts.MCTSNode.__init__


def _init(self, parent, prior_p, board_state):
    _real_init(self, parent, prior_p, board_state)
    NODES.append(self)


mcts.MCTSNode.__init__ = _init

_real_expand = mcts.MCTSNode.expand


def _expand(self, policy_probs, legal_moves):
    k = PROB_HASH.get(hashlib.sha1(np.ascontiguousarray(policy_probs).tobytes()).hexdigest())
    if k is not None:
        EVALS[k][3].update(utils.move_to_index(m) for m in legal_moves)
    return _real_expand(self, policy_probs, legal_moves)


mcts.MCTSNode.expand = _expand

BATCHES = []
_real_eval_batch = mcts._evaluate_batch


def _eval_batch(nodes, paths, model):
    BATCHES.append([len(nodes), len({id(n) for n in nodes})])
    return _real_eval_batch(nodes, paths, model)


mcts._evaluate_batch = _eval_batch


def f32bits(x) -> int:
    return int(np.float32(x).view(np.uint32))


def canonical_tree(root):
    out = {}

    def walk(node, path):
        out["/".join(path)] = [int(node.n_visits), None if not path else f32bits(node.q_value),
                               f32bits(node.prior_p), len(node.children)]
        for mv, ch in node.children.items():
            walk(ch, path + [mv.uci()])

    walk(root, [])
    return out


def set_config(cfg):
    config.NUM_SIMULATIONS = cfg.get("num_simulations", 250)
    config.MCTS_BATCH_SIZE = cfg.get("batch_size", 96)
    config.CPUCT = cfg.get("cpuct", 1.0)
    config.WIDEN_COEFF = cfg.get("widen_coeff", 1.5)
    config.DIRICHLET_ALPHA = cfg.get("dirichlet_alpha", 0.1)
    config.DIRICHLET_EPSILON = cfg.get("dirichlet_eps", 0.25)
    config.MAX_GAME_MOVES = cfg.get("max_game_moves", 16384)


def build_context(fen, moves, uci_style=False):
    """What self_play.py:91-96,171-184 (or uci.py:161-199) builds before calling run_mcts."""
    board = chess.Board(fen)
    tracker = utils.RepetitionTracker()
    tracker.add_board(board)
    hist = [board.copy()]
    for u in moves:
        board.push(chess.Move.from_uci(u))
        tracker.add_board(board)
        hist.append(board.copy())
    if uci_style:  # uci.py:199 history[-8:] incl. the current board, then uci.py:62 history[-7:]
        history = hist[-8:]
        history = history[max(0, len(history) - 7):]
    else:  # self_play.py:109
        history = hist[max(0, len(hist) - 8):-1]
    return board, history, tracker


def run_search_case(case):
    set_config(case["config"])
    model = RecordingModel(FakeNet(scale=case["scale"], salt=case["salt"]))
    board, history, tracker = build_context(case["fen"], case["moves"], case.get("uci_style", False))
    np.random.seed(case["seed"])
    random.seed(case["seed"])
    NODES.clear()
    BATCHES.clear()
    exp = {}
    try:
        best, pi = mcts.run_mcts(board, model, history, tracker)
    except ValueError as e:
        exp["raises"] = "ValueError"
        return exp
    root = NODES[0]
    nz = np.nonzero(pi)[0]
    exp.update(best=best.uci(), pi=[[int(i), f32bits(pi[i])] for i in nz], tree=canonical_tree(root),
               batches=[list(b) for b in BATCHES], n_nodes=len(canonical_tree(root)))
    return exp


def run_game_case(case):
    set_config(case["config"])
    model = RecordingModel(FakeNet(scale=case["scale"], salt=case["salt"]))
    np.random.seed(case["seed"])
    random.seed(case["seed"])
    # self_play.run_self_play_game always starts from chess.Board() (self_play.py:91); fixtures that
    # want another start position swap the constructor default for the duration of the call.
    real_board = chess.Board
    if case.get("fen"):
        fen = case["fen"]

        class _B(real_board):
            def __init__(self, f=fen, **k):
                super().__init__(f, **k)

        chess.Board = _B
    stacks = []
    real_add = utils.RepetitionTracker.add_board

    def _add(self, board):  # called after every real move (self_play.py:93,182)
        stacks.append([m.uci() for m in board.move_stack])
        return real_add(self, board)

    utils.RepetitionTracker.add_board = _add
    try:
        data = self_play.run_self_play_game(model, 0)
    finally:
        chess.Board = real_board
        utils.RepetitionTracker.add_board = real_add
    if data is None:
        return {"aborted": True}
    exp = {"moves": stacks[-1], "n_records": len(data), "z": [float(z) for _, _, z in data],
           "z_signbit": [bool(np.signbit(z)) for _, _, z in data],
           "pi": [[[int(i), f32bits(p[i])] for i in np.nonzero(p)[0]] for _, p, _ in data],
           "state_sha1": [planes_key(s.numpy()) for s, _, _ in data]}
    return exp


SEARCH_CASES = [
    dict(name="startpos_250", fen=chess.STARTING_FEN, moves=[], seed=1, scale=6.0, salt=0,
         config=dict(num_simulations=250)),
    dict(name="startpos_50_cfg1", fen=chess.STARTING_FEN, moves=[], seed=0, scale=6.0, salt=1,
         config=dict(num_simulations=50)),
    dict(name="ruy_10plies_400", fen=chess.STARTING_FEN,
         moves="e2e4 e7e5 g1f3 b8c6 f1b5 a7a6 b5a4 g8f6 e1g1 f8e7".split(), seed=2, scale=6.0, salt=2,
         config=dict(num_simulations=400)),
    dict(name="kiwipete_250", fen="r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", moves=[],
         seed=3, scale=6.0, salt=3, config=dict(num_simulations=250)),
    dict(name="promotion_250", fen="rnbqkbnr/pppp1Ppp/8/8/8/8/PPPP1PPP/RNBQKBNR w KQkq - 0 1", moves=[], seed=4,
         scale=6.0, salt=4, config=dict(num_simulations=250)),
    dict(name="uniform_priors_ties", fen=chess.STARTING_FEN, moves="d2d4 d7d5".split(), seed=5, scale=0.0, salt=5,
         config=dict(num_simulations=250)),
    dict(name="small_batch_deep", fen=chess.STARTING_FEN, moves="c2c4 e7e5 b1c3".split(), seed=6, scale=8.0, salt=6,
         config=dict(num_simulations=200, batch_size=8)),
    dict(name="in_check_evasions", fen="rnbqkbnr/ppp2ppp/8/1B1pp3/4P3/8/PPPP1PPP/RNBQK1NR b KQkq - 1 3", moves=[],
         seed=7, scale=6.0, salt=7, config=dict(num_simulations=150)),
    dict(name="en_passant_root", fen="rnbqkbnr/ppp1pppp/8/8/3pP3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 3", moves=[],
         seed=8, scale=6.0, salt=8, config=dict(num_simulations=150)),
    dict(name="repetition_children_terminal", fen=chess.STARTING_FEN,
         moves="g1f3 g8f6 f3g1 f6g8 g1f3 g8f6".split(), seed=9, scale=6.0, salt=9,
         config=dict(num_simulations=250)),
    dict(name="halfmove_98_all_children_claimable", fen="8/8/4k3/8/8/3K4/8/6R1 w - - 98 80", moves=[], seed=10,
         scale=6.0, salt=10, config=dict(num_simulations=120)),
    dict(name="root_terminal_claimable", fen="8/8/4k3/8/8/3K4/8/6R1 w - - 99 80", moves=[], seed=11, scale=6.0,
         salt=11, config=dict(num_simulations=40)),
    dict(name="uci_style_history", fen=chess.STARTING_FEN, moves="e2e4 c7c5 g1f3".split(), seed=12, scale=6.0,
         salt=12, uci_style=True, config=dict(num_simulations=100)),
    dict(name="no_dirichlet_800", fen=chess.STARTING_FEN, moves="e2e4".split(), seed=13, scale=6.0, salt=13,
         config=dict(num_simulations=800, dirichlet_alpha=0.0)),
    dict(name="cpuct_widen_variants", fen=chess.STARTING_FEN, moves="g1f3".split(), seed=14, scale=6.0, salt=14,
         config=dict(num_simulations=300, cpuct=2.5, widen_coeff=2.0, batch_size=32)),
]

# cases whose salt is searched for so that simulations hit terminal leaves (mate / draw)
TERMINAL_HUNT = [
    dict(name="mate_in_one_hit", fen="k7/8/1K6/8/8/8/8/7R w - - 0 1", moves=[], seed=20, scale=6.0,
         config=dict(num_simulations=150), want="mate"),
    dict(name="capture_to_bare_kings", fen="8/8/8/8/8/2k5/1p6/K7 w - - 0 1", moves=[], seed=21, scale=6.0,
         config=dict(num_simulations=150), want="draw"),
    dict(name="checkmated_root_raises", fen="k6R/8/1K6/8/8/8/8/8 b - - 1 1", moves=[], seed=22, scale=6.0, salt=0,
         config=dict(num_simulations=10), want=None),
]

GAME_CASES = [
    dict(name="game_cfg1_16plies", seed=0, scale=6.0, salt=100, config=dict(num_simulations=50, max_game_moves=16)),
    dict(name="game_endgame_clock90", fen="k7/8/1K6/8/8/8/8/7R w - - 90 60", seed=1, scale=6.0, salt=101,
         config=dict(num_simulations=40, batch_size=16)),
    dict(name="game_temp_final", fen="r1bqkbnr/pppp1ppp/2n5/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R w KQkq - 2 29", seed=2,
         scale=6.0, salt=102, config=dict(num_simulations=30, max_game_moves=6)),
]


def gen_g2():
    searches = []
    for case in SEARCH_CASES:
        exp = run_search_case(case)
        searches.append(dict(case=case, expect=exp))
        print(f"[g2] {case['name']}: nodes={exp.get('n_nodes')} batches={exp.get('batches')} best={exp.get('best')}")
    for case in TERMINAL_HUNT:
        if case["want"] is None:
            exp = run_search_case(case)
            searches.append(dict(case=case, expect=exp))
            print(f"[g2] {case['name']}: {exp}")
            continue
        found = False
        for salt in range(200, 400):
            c = dict(case, salt=salt)
            before = set(EVALS)
            exp = run_search_case(c)
            rows = sum(b[0] for b in exp["batches"])
            term_sims = c["config"]["num_simulations"] - rows
            if term_sims > 0:
                c = {k: v for k, v in c.items() if k != "want"}
                exp["n_terminal_sims"] = term_sims
                searches.append(dict(case=c, expect=exp))
                print(f"[g2] {case['name']}: salt={salt} terminal_sims={term_sims} batches={exp['batches']}")
                found = True
                break
            for k in set(EVALS) - before:  # drop evals of rejected salts
                del EVALS[k]
        assert found, case["name"]
    games = []
    for case in GAME_CASES:
        exp = run_game_case(case)
        games.append(dict(case=case, expect=exp))
        print(f"[g2] {case['name']}: records={exp.get('n_records')} z={exp.get('z', [])[:4]}")
    full = sorted(k for k in EVALS if EVALS[k][2])
    sparse = sorted(k for k in EVALS if not EVALS[k][2])
    ptr, idx, val = [0], [], []
    for k in sparse:
        ii = np.array(sorted(EVALS[k][3]), dtype=np.int32)
        idx.append(ii)
        val.append(EVALS[k][0][ii])
        ptr.append(ptr[-1] + len(ii))
    np.savez(os.path.join(OUT, "g2_evals.npz"),
             full_keys=np.array(full), full_probs=np.stack([EVALS[k][0] for k in full]),
             full_values=np.array([EVALS[k][1] for k in full], np.float32),
             sparse_keys=np.array(sparse), sparse_ptr=np.array(ptr, np.int64),
             sparse_idx=np.concatenate(idx), sparse_val=np.concatenate(val),
             sparse_values=np.array([EVALS[k][1] for k in sparse], np.float32))
    keys = full + sparse
    json.dump(searches, open(os.path.join(OUT, "g2_searches.json"), "w"), indent=0)
    json.dump(games, open(os.path.join(OUT, "g2_games.json"), "w"), indent=0)
    print(f"[g2] {len(keys)} evals, {len(searches)} searches, {len(games)} games")


NET_SIZES = {"3+1x64": (3, 1, 64), "8+2x128": (8, 2, 128), "15+5x256": (15, 5, 256)}
G1_FENS = [chess.STARTING_FEN, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
           "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 b - - 17 42"]


def gen_g1():
    torch.softmax = _real_softmax
    out = {}
    xs = []
    for fen in G1_FENS:
        b = chess.Board(fen)
        t = utils.RepetitionTracker()
        t.add_board(b)
        xs.append(utils.encode_board(b, [b], t))
    x = torch.stack(xs)
    out["inputs"] = x.numpy()
    for name, (rb, se, f) in NET_SIZES.items():
        config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = rb, se, f
        net = network.PolicyValueNet().eval()
        hash_init_(net)
        with torch.no_grad():
            logits, value = net(x)
        out[f"logits_{name}"] = logits.numpy()
        out[f"value_{name}"] = value.numpy()
        out[f"nparams_{name}"] = np.array(sum(p.numel() for p in net.parameters()))
        out[f"nkeys_{name}"] = np.array(len(net.state_dict()))
        print(f"[g1] {name}: params={int(out[f'nparams_{name}'])} keys={int(out[f'nkeys_{name}'])} "
              f"logit range [{logits.min():.3f},{logits.max():.3f}] value {value.flatten().tolist()}")
    np.savez_compressed(os.path.join(OUT, "g1_net.npz"), **out)


def gen_g4():
    import contextlib
    import io

    out = []
    for fen in [chess.STARTING_FEN, "rnbqkbnr/pppp1Ppp/8/8/8/8/PPPP1PPP/RNBQKBNR w KQkq - 0 1",
                "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
                "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R b KQkq - 0 1",
                "4k3/P6P/8/8/8/8/p6p/4K3 b - - 0 1"]:
        b = chess.Board(fen)
        with contextlib.redirect_stdout(io.StringIO()):
            errors = utils.test_move_indexing(b)  # utils.py:399-464
        table = [[m.uci(), utils.move_to_index(m)] for m in b.legal_moves]
        out.append(dict(fen=fen, errors=int(errors), moves=table))
        print(f"[g4] {fen}: errors={errors} moves={len(table)}")
    json.dump(out, open(os.path.join(OUT, "g4_codec.json"), "w"), indent=0)


# Full-length traces ("to the end or ..."): the unmodified reference plays
#   (a) one game from the start position at 50 simulations per move to natural termination or 120 plies -- the temperature switch at
#       fullmove 30 (self_play.py:66) happens inside it, the end-of-game tracker (E8) covers a long game;
#   (b) one game from fullmove 28 with the fifty-move clock at 60: it crosses the temperature threshold after four plies and ends BY RULE
#       (mate, or the claimable draw of is_game_over(claim_draw=True), self_play.py:101-102,190-208).
# Written to g5_long_games.json / g5_long_evals.npz so that the round-1 fixtures stay byte-identical.
LONG_GAME_CASES = [
    dict(name="game_long_startpos_120", seed=5, scale=6.0, salt=300, config=dict(num_simulations=50, max_game_moves=120)),
    dict(name="game_long_threshold_to_rule", fen="8/5k2/8/8/3K4/8/8/R7 w - - 60 28", seed=6, scale=6.0, salt=301,
         config=dict(num_simulations=50)),
]


def gen_g5():
    EVALS.clear()
    PROB_HASH.clear()
    games = []
    for case in LONG_GAME_CASES:
        exp = run_game_case(case)
        games.append(dict(case=case, expect=exp))
        print(f"[g5] {case['name']}: records={exp.get('n_records')} last z={exp.get('z', [])[-2:]} moves={' '.join(exp.get('moves', [])[-6:])}")
    full = sorted(k for k in EVALS if EVALS[k][2])
    sparse = sorted(k for k in EVALS if not EVALS[k][2])
    ptr, idx, val = [0], [], []
    for k in sparse:
        ii = np.array(sorted(EVALS[k][3]), dtype=np.int32)
        idx.append(ii)
        val.append(EVALS[k][0][ii])
        ptr.append(ptr[-1] + len(ii))
    np.savez_compressed(os.path.join(OUT, "g5_long_evals.npz"),
                        full_keys=np.array(full), full_probs=np.stack([EVALS[k][0] for k in full]),
                        full_values=np.array([EVALS[k][1] for k in full], np.float32),
                        sparse_keys=np.array(sparse), sparse_ptr=np.array(ptr, np.int64),
                        sparse_idx=np.concatenate(idx), sparse_val=np.concatenate(val),
                        sparse_values=np.array([EVALS[k][1] for k in sparse], np.float32))
    json.dump(games, open(os.path.join(OUT, "g5_long_games.json"), "w"), indent=0)
    print(f"[g5] {len(full) + len(sparse)} evals, {len(games)} games")


if __name__ == "__main__":
    if "--long" in sys.argv:  # only the full-length games (the other fixtures are left as they are)
        gen_g5()
        sys.exit(0)
    gen_g4()
    gen_g2()
    gen_g1()
    gen_g5()
