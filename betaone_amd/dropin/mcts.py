"""
mcts -- `run_mcts(root_board, model, history, tracker)` of /root/reference/mcts.py:155-280 over the
MI355X engine.  Same signature, same return value `(best_move, pi float32[4672])`, same error
behaviour (ValueError on a root without legal moves, mcts.py:279), callable from uci.py's search
thread (ctypes releases the GIL).  The tree arithmetic of MCTSNode (mcts.py:19-152) runs in
betaone_amd/csrc/bo_tree.h; this file only marshals the caller's duck-typed python-chess objects
into plain data and runs the evaluate stage under PyTorch-ROCm.
"""
import sys
import threading
from typing import List, Optional, Tuple

import numpy as np
import torch

import config
from betaone_amd import engine as E

_lock = threading.Lock()
# Interruptible search.  The reference polls its stop flag only between whole searches
# (uci.py:73); here a running search is interrupted between two evaluate->step replays: set `stop_event` (or call
# request_stop()) from another thread and run_mcts returns the result of the simulations completed so far -- exactly what
# the reference returns for that NUM_SIMULATIONS.  uci.py keeps its own module-level `stop_event` (uci.py:44) and runs as
# __main__: that event is honoured too, so the unchanged uci.py stops mid-search.  The caller clears the event.
stop_event = threading.Event()
RUN_AHEAD = 4
last_search = {"simulations": 0, "stopped": False}   # of the most recent run_mcts call


def request_stop():
    stop_event.set()


def _stop_requested() -> bool:
    if stop_event.is_set():
        return True
    ev = getattr(sys.modules.get("__main__"), "stop_event", None)
    return isinstance(ev, threading.Event) and ev.is_set()
_ctx = {}            # config tuple -> (Engine, nn_in tensor)
_fast = {}           # id(model) -> [parameter versions, tuned inference copy, captured hipGraph step]


def _context():
    device = E.runtime_device(config.DEVICE)  # 'cuda' -> cuda:<current device>; raises off the GPU (no CPU path)
    key = (config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.CPUCT, config.WIDEN_COEFF, config.DIRICHLET_ALPHA,
           config.DIRICHLET_EPSILON, str(device), id(E.load_hip_library()))
    if key not in _ctx:
        eng = E.Engine(1, num_simulations=config.NUM_SIMULATIONS, mcts_batch_size=config.MCTS_BATCH_SIZE,
                       cpuct=config.CPUCT, widen_coeff=config.WIDEN_COEFF, dirichlet_alpha=config.DIRICHLET_ALPHA,
                       dirichlet_epsilon=config.DIRICHLET_EPSILON, max_plies=config.ENGINE_MAX_PLIES or 16386,
                       device=device.index if device.index is not None else 0)
        nn_in = torch.zeros((1, config.INPUT_CHANNELS, 8, 8), dtype=torch.float32, device=device)
        _ctx.clear()  # one live engine is enough for the single-search path
        _ctx[key] = (eng, nn_in)
    return _ctx[key]


def _castling_bits(rights: int) -> int:
    return (1 if rights & (1 << 7) else 0) | (2 if rights & 1 else 0) | (4 if rights & (1 << 63) else 0) | \
           (8 if rights & (1 << 56) else 0)


def board_to_position(board) -> E.BoPosition:
    p = E.BoPosition()
    occ = board.occupied_co
    for i, v in enumerate((board.pawns, board.knights, board.bishops, board.rooks, board.queens, board.kings,
                           occ[True], occ[False])):
        p.bb[i] = int(v)
    p.turn = 1 if board.turn else 0
    rights = board.clean_castling_rights() if hasattr(board, "clean_castling_rights") else board.castling_rights
    p.castling = _castling_bits(int(rights))
    p.ep_square = -1 if board.ep_square is None else int(board.ep_square)
    p.ep_key = -2
    p.halfmove_clock, p.fullmove_number = int(board.halfmove_clock), int(board.fullmove_number)
    return p


def key_to_position(key) -> E.BoPosition:
    """python-chess Board._transposition_key() tuple -> plain data."""
    p = E.BoPosition()
    for i in range(8):
        p.bb[i] = int(key[i])
    p.turn = 1 if key[8] else 0
    p.castling = _castling_bits(int(key[9]))
    p.ep_square = p.ep_key = -1 if key[10] is None else int(key[10])
    p.halfmove_clock, p.fullmove_number = 0, 1
    return p


def board_to_stack(board) -> Tuple[str, Optional[str]]:
    """(FEN of the bottom of the move stack, UCI moves) -- the draw rules need the whole stack (mcts.py:36,152)."""
    b = board.copy()
    ucis = []
    while b.move_stack:
        ucis.append(b.pop().uci())
    try:
        fen = b.fen(en_passant="fen")
    except TypeError:
        fen = b.fen()
    return fen, (" ".join(reversed(ucis)) or None)


def _make_move(board, m: int):
    f, t, promo = m & 63, (m >> 6) & 63, (m >> 12) & 7
    mod = sys.modules.get(type(board).__module__)
    cls = getattr(mod, "Move", None)
    if cls is None:
        import chess

        cls = chess.Move
    return cls(f, t, promotion=promo or None)


def _policy_kind() -> int:
    return E.POLICY_PROBS if config.POLICY_SOFTMAX == "torch" else E.POLICY_LOGITS


def _evaluate(model, nn_in):
    """-> (policy, value): the policy is softmax probabilities (mcts.py:185,287) or raw logits, see config.POLICY_SOFTMAX."""
    with torch.no_grad():
        if config.AUTOCAST:
            with torch.autocast(nn_in.device.type):
                logits, value = model(nn_in)
        else:
            logits, value = model(nn_in)
        logits = logits.float()
        if _policy_kind() == E.POLICY_PROBS:
            logits = torch.softmax(logits, dim=1)
    return logits.contiguous(), value.float().contiguous()


class _GraphStep:
    """`net forward -> bo_step` for the single-search path, captured once per (model, engine) as a hipGraph:
    uci.py calls run_mcts back to back on one position (uci.py:72-93), 1 + ceil(sims/96) batch-1 evaluations each."""

    def __init__(self, model, eng, nn_in):
        self.model, self.eng, self.nn_in, self.kind = model, eng, nn_in, _policy_kind()
        dev = nn_in.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                _evaluate(model, nn_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the evaluate stage's own fault words (the one-launch tower's hand-off timeout / saturation, the split tower's saturation)
        # come back with every result block this engine fetches: run_mcts checks them once per search, in the round trip it makes anyway
        inner = getattr(model, "net", model)
        words = getattr(inner, "overflow_words", None)
        eng.watch(*(words() if words is not None else (0, 1)))
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):  # (other threads -- uci.py's stdin loop, a process group's watchdog -- are not held to capture-safe calls)
            self.out = _evaluate(model, nn_in)
            eng.step(self.out[0].data_ptr(), self.out[1].data_ptr(), self.kind, nn_in.data_ptr(),
                     torch.cuda.current_stream(dev).cuda_stream)

    def __call__(self):
        self.graph.replay()


def _fast_path(model, eng, nn_in):
    """Tuned inference copy (BN folded, fused epilogues when fp32) + captured step; rebuilt when the caller's
    parameters change (their version counters move on load_state_dict / optimizer steps)."""
    if nn_in.device.type != "cuda" or config.AUTOCAST or not hasattr(model, "for_inference"):
        return None
    ver = tuple(p._version for p in model.parameters())
    hit = _fast.get(id(model))
    if hit is None or hit[0] != ver:  # new weights: tune the inference copy again (times the candidate kernels, ~1 s)
        from betaone_amd.nn_tune import best_inference_copy

        _fast.clear()
        hit = _fast[id(model)] = [ver, best_inference_copy(model, 1, nn_in.device, next(model.parameters()).dtype), None]
    if hit[2] is None or hit[2].eng is not eng or hit[2].kind != _policy_kind():  # new search settings: only re-capture
        hit[2] = _GraphStep(hit[1], eng, nn_in)
    return hit[2]


def run_mcts(root_board, model, history: List, tracker) -> Tuple[object, np.ndarray]:
    with _lock:
        eng, nn_in = _context()
        fen, moves = board_to_stack(root_board)
        hist = [board_to_position(b) for b in list(history)[-7:]]          # (history + [root])[-8:], mcts.py:180
        trk = [(key_to_position(k), int(c)) for k, c in tracker.counts.items() if c > 0]
        eng.reset_ex([0], [fen], [moves], [hist], [trk])
        n_legal, terminal, _ = eng.root_info()
        noise = None
        if config.DIRICHLET_ALPHA > 0:
            noise = np.zeros((1, E.MAX_LEGAL))
            if terminal[0] == 0:
                noise[0, :n_legal[0]] = np.random.dirichlet([config.DIRICHLET_ALPHA] * int(n_legal[0]))  # mcts.py:192
        cuda = nn_in.device.type == "cuda"
        stream = torch.cuda.current_stream(nn_in.device).cuda_stream if cuda else 0
        eng.search_begin([1], noise, nn_in.data_ptr(), stream)
        eng.step(0, 0, E.POLICY_NONE, nn_in.data_ptr(), stream)
        graph_step = _fast_path(model, eng, nn_in)
        keep = None
        burst = 1 + -(-config.NUM_SIMULATIONS // config.MCTS_BATCH_SIZE) if (graph_step is not None and terminal[0] == 0) else 0
        stopped = False
        ahead = []  # completion events of the replays in flight: the host stays at most RUN_AHEAD replays in front of the GPU,
        while not stopped:  # so a stop request takes effect within that many evaluations (and the queue never runs dry)
            for _ in range(burst):  # the expected number of evaluations without a host round trip in between
                if _stop_requested():
                    stopped = True
                    break
                graph_step()
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(nn_in.device))
                ahead.append(ev)
                if len(ahead) > RUN_AHEAD:
                    ahead.pop(0).synchronize()
            burst = 0
            if stopped:
                break
            running, _, _ = eng.poll(stream, want_mask=False)
            if running == 0:
                break
            if _stop_requested():
                stopped = True
            elif graph_step is not None:
                graph_step()
            else:
                keep = _evaluate(model, nn_in)
                eng.step(keep[0].data_ptr(), keep[1].data_ptr(), _policy_kind(), nn_in.data_ptr(), stream)
        sims_done = config.NUM_SIMULATIONS
        if stopped:  # tail flush + result of the simulations completed so far (mcts.py:256-280 at that NUM_SIMULATIONS)
            sims_done = int(eng.search_stop(None, stream)[0])
        last_search.update(simulations=sims_done, stopped=stopped)
        eng.check_status()
        res = eng.result(stream)
        if graph_step is not None and eng.watch_seen():  # an evaluation of this search was invalid: say which and why instead of returning a move
            inner = getattr(graph_step.model, "net", graph_step.model)
            torch.cuda.synchronize(nn_in.device)
            getattr(inner, "check_overflow", lambda: None)()  # (raises with the reason; re-arms the stage's counters)
            raise E.EngineError("run_mcts: the evaluate stage reported a fault during this search")
        if res["best_idx"][0] < 0:
            raise ValueError("max() arg is an empty sequence")  # mcts.py:279 on a root without legal moves
        pi = np.zeros(config.NUM_ACTIONS, dtype=np.float32)
        n = int(res["n"][0])
        pi[res["idx"][0, :n]] = res["val"][0, :n]
        return _make_move(root_board, int(res["best_move"][0])), pi


class MCTS:
    """North-star alias (BASELINE.json): MCTS(model).search(board, history, tracker) == run_mcts(...)."""

    def __init__(self, model):
        self.model = model

    def search(self, root_board, history, tracker):
        return run_mcts(root_board, self.model, history, tracker)
