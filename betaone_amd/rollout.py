"""
betaone_amd/rollout.py -- the self-play rollout driver: G concurrent games on one MI355X.

Host side of what the reference does one game at a time in Python
  run_mcts              /root/reference/mcts.py:155-280
  _evaluate_batch       /root/reference/mcts.py:283-295
  run_self_play_game    /root/reference/self_play.py:84-216
re-organised for the GPU: the trees of all G games live in HBM and advance in lock step through
the engine's kernels (include/betaone_engine.h); each step evaluates ONE leaf per game, so the net
always sees a static [G,120,8,8] batch (NN row g <-> game slot g) and `net forward + tree step` is
captured once as a hipGraph and replayed.  The host only does the per-move NumPy work whose RNG
stream has to stay bit-identical to the reference (Dirichlet noise, temperature sampling) and
recycles the slots of finished games.

There is no CPU path here: the engine is csrc/libbetaone_hip.so or an exception.
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import engine as E
from . import sampling

# hipGraph capture mode of every capture this package makes.  "thread_local": only the CAPTURING thread is held to capture-safe calls.
# The default ("global") forbids capture-unsafe HIP calls in EVERY thread while a capture is open -- and torch.distributed's RCCL
# watchdog thread polls its collectives' events (hipEventQuery) at any time: with a record exchange in flight, or right after the
# bench's communicator census, a capture in global mode dies with "operation not permitted when stream is capturing" (found by the
# N = 1 RCCL rehearsal of round 5; the N > 1 run would have hit it on every rank).
CAPTURE_MODE = "thread_local"


@dataclass
class GameState:
    """Host-side bookkeeping of one game slot (what run_self_play_game keeps in locals)."""
    game_id: int
    rng: object                      # numpy RandomState (or the numpy.random module)
    start_fen: Optional[str] = None
    start_fullmove: int = 1
    start_white: bool = True
    plies: int = 0                   # moves on the game's stack (incl. a prefix handed to start_games)
    first_ply: int = 0               # length of that prefix: the first ply searched here
    pis: List = field(default_factory=list)   # per searched ply: (action indices, float32 probabilities)

    def fullmove_number(self) -> int:          # board.fullmove_number of the current root (self_play.py:104)
        return self.start_fullmove + (self.plies + (0 if self.start_white else 1)) // 2


class SparsePis:
    """The sparse pi of every searched ply of one game, as a sequence of (action indices, float32 probabilities) backed by
    three arrays (entries per ply n[T], idx[T,K], val[T,K]) -- a finished game is handed over without a Python loop over
    its plies (the reference's search leaves at most two non-zero entries per pi)."""

    def __init__(self, n: np.ndarray, idx: np.ndarray, val: np.ndarray):
        self.n, self.idx, self.val = n, idx, val

    def __len__(self) -> int:
        return len(self.n)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return SparsePis(self.n[i], self.idx[i], self.val[i])
        k = int(self.n[i])
        return self.idx[i, :k], self.val[i, :k]

    def __iter__(self):
        return (self[i] for i in range(len(self.n)))

    def flat(self):
        """(ptr[T+1], idx[total], val[total]): the CSR form the record wire format stores."""
        mask = np.arange(self.idx.shape[1])[None, :] < self.n[:, None]
        ptr = np.zeros(len(self.n) + 1, dtype=np.int32)
        np.cumsum(self.n, out=ptr[1:])
        return ptr, self.idx[mask].astype(np.int32, copy=False), self.val[mask].astype(np.float32, copy=False)


@dataclass
class FinishedGame:
    game_id: int
    slot: int
    moves: List[int]                 # from|to<<6|promo<<12
    positions: List[E.BoPosition]    # positions[i] = position before moves[i]; last = final position
    pis: List                        # sparse pi of ply first_ply + i
    outcome: float                   # utils.get_game_outcome of the final board, 0.0 if not over (self_play.py:190-197)
    terminal: int                    # 0 stopped by move limit, 1 checkmate, 2 draw
    first_ply: int = 0               # plies before it were handed to start_games(moves=...), not searched here

    def z(self, i: int) -> float:    # self_play.py:202, for record i (ply first_ply + i)
        return self.outcome if self.positions[self.first_ply + i].turn == 1 else -self.outcome


def _fen_meta(fen: Optional[str]):
    if not fen:
        return 1, True
    parts = fen.split()
    white = len(parts) < 2 or parts[1] == "w"
    full = max(1, int(parts[5])) if len(parts) > 5 else 1
    return full, white


def _on_main(fn):
    """Entry points of a Rollout that was given its own stream issue everything they launch to that stream."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if self._main is None:
            return fn(self, *a, **kw)
        with torch.cuda.stream(self._main):
            return fn(self, *a, **kw)
    return wrapped


class Rollout:
    STAMP_RING = None        # LAB (BO_STAMPS=1, scripts/cohort_timeline.py): a device ring every Rollout's phases are stamped into
    STAMP_CAP = 1 << 18
    _n_instances = 0

    def _stamp(self, phase: int):
        if Rollout.STAMP_RING is not None:
            self.eng.lib.bo_debug_stamp(Rollout.STAMP_RING.data_ptr(), self._stamp_id * 16 + phase, Rollout.STAMP_CAP, self._stream())

    def __init__(self, model: torch.nn.Module, n_games: int, *, num_simulations: int = 250, mcts_batch_size: int = 96,
                 cpuct: float = 1.0, widen_coeff: float = 1.5, dirichlet_alpha: float = 0.1,
                 dirichlet_epsilon: float = 0.25, max_plies: Optional[int] = None, max_game_moves: int = 16384,
                 temperature=(30, 1.0, 0.1), device: str = "cuda:0", use_graph: bool = True, autocast: bool = False,
                 rng_mode: str = "python", policy_kind: str = "logits", fast: bool = False, leaves_per_step: int = 16,
                 fast_arena_granules: int = 0, stream: Optional["torch.cuda.Stream"] = None, time_tower: bool = False):
        self.device = E.runtime_device(device)
        # `stream`: every launch of this Rollout goes to that HIP stream (CohortRollout: one stream per cohort of games, so that
        # one cohort's tower runs while another's tree step / head kernels / host turn are in progress); None = torch's current one
        self._main = stream
        self._stamp_id = Rollout._n_instances
        Rollout._n_instances += 1
        # time_tower (measurement, bench.py): every launch of the split-precision tower made by THIS Rollout notes its own duration in
        # this buffer (bo_nn_tower_forward_timed: works inside captured graphs) -- [seq | arrivals | start[4096] | end[4096]]
        self.tower_timing = torch.zeros(2 + 2 * 4096, dtype=torch.int64, device=self.device) if time_tower else None  # 'cuda' -> cuda:<current device>: engine, NN rows and model on ONE GPU
        self.G = int(n_games)
        self.S, self.B = int(num_simulations), int(mcts_batch_size)
        self.alpha = float(dirichlet_alpha)
        self.max_game_moves = int(max_game_moves)
        if max_plies is None:  # room for the longest game the caller allows (self_play.py:102; config.MAX_GAME_MOVES = 16384)
            max_plies = min(self.max_game_moves, 16384) + 2
        self.temperature = temperature
        self.model = model
        self.autocast = autocast
        dev_index = self.device.index if self.device.index is not None else 0
        self.eng = E.Engine(self.G, num_simulations=self.S, mcts_batch_size=self.B, cpuct=cpuct, widen_coeff=widen_coeff,
                            dirichlet_alpha=dirichlet_alpha, dirichlet_epsilon=dirichlet_epsilon, max_plies=max_plies,
                            device=dev_index, fast=fast, leaves_per_step=leaves_per_step, fast_arena_granules=fast_arena_granules)
        # fast=True: csrc/bo_fast.h (virtual loss, L leaves per game per step) -- NOT the reference's search semantics
        self.fast, self.L = bool(fast), self.eng.L
        self.nn_in = torch.zeros((self.G * self.L, E.INPUT_CHANNELS, 8, 8), dtype=torch.float32, device=self.device)
        # evaluations per search before the first poll: reference semantics = root + one per batch; fast mode = one per L
        # simulations (a root kept from the previous search needs no evaluation of its own; a fresh one costs one more round)
        self.expected_evals = math.ceil(self.S / self.L) if self.fast else 1 + math.ceil(self.S / self.B)
        self.games: List[Optional[GameState]] = [None] * self.G
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self._graph = None
        self._graphs_n = {}         # n -> (graph of n back-to-back evaluate->step iterations, its output tensors)
        self._logits = self._value = None
        self.n_forward = 0          # NN forwards issued
        self.n_sims = 0             # simulations completed (NUM_SIMULATIONS per finished search)
        self.n_plies = 0
        self.host_seconds = 0.0     # time spent in per-move host work (noise + sampling + bookkeeping)
        # rng_mode "python": one numpy RandomState (or the numpy.random module) per game, drawn in Python.
        # rng_mode "native": the same legacy MT19937 streams kept inside the engine (bo_hostrng.h); `rngs` passed to
        # start_games are then integer seeds, and a ply costs three library calls instead of a Python loop over games.
        assert rng_mode in ("python", "native")
        self.rng_mode = rng_mode
        # policy_kind "logits": the engine's own in-kernel softmax (one pass over the row);
        # "probs": torch.softmax(logits, dim=1) exactly as mcts.py:185,287 and the engine gathers probabilities.
        assert policy_kind in ("logits", "probs")
        self.policy_kind = E.POLICY_LOGITS if policy_kind == "logits" else E.POLICY_PROBS
        G = self.G
        self._fgraph = None
        self._side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None  # finished games' hand-over beside the search
        self._noise_pending = False
        self._fwd_early = False  # the next ply's root evaluation has been enqueued already (behind bo_selfplay_turn)
        import os as _os
        self.ply_profile = [] if _os.environ.get("BO_PLY_PROFILE", "0") not in ("", "0") else None  # [(phase, seconds)] of the native ply path
        self._t_ret = None
        self._begun = None          # (n_legal, terminal, go) of searches already begun by the previous selfplay_turn
        self._turn_due = None       # go[G] of a ply whose searches ply_begin enqueued and ply_end has not turned yet
        self._ply_event = None      # recorded behind everything ply_begin enqueued (ply_ready)
        self._prefetched = False    # the result block of the enqueued searches is on its way to pinned host memory (_prefetch_result)
        self._auto = False          # the due ply's TURN is enqueued on the device behind its searches (_enqueue_autoturn): ply_end only collects it
        self.device_turn = os.environ.get("BETAONE_DEVICE_TURN", "1") != "0"  # (0: the host samples and plays, for A/B runs)
        self._begun_want = None
        self._active = np.zeros(G, dtype=bool)
        self._plies = np.zeros(G, dtype=np.int64)
        self._start_full = np.ones(G, dtype=np.int64)
        self._start_black = np.zeros(G, dtype=np.int64)
        self._start_step = np.zeros(G, dtype=np.int64)
        self._first_ply = np.zeros(G, dtype=np.int64)
        self._step = 0
        self._hist = {}             # step -> (go[G], n[G], idx[G,K], val[G,K]) sparse pi of every game that searched at that step
        # reference semantics: pi has <= 2 entries, kept per slot and ply in dense arrays that grow with the longest game
        self._pk = 0 if self.fast else 2
        self._pi_n = np.zeros((G, 256), np.int32)
        self._pi_idx = np.zeros((G, 256, max(1, self._pk)), np.int32)
        self._pi_val = np.zeros((G, 256, max(1, self._pk)), np.float32)
        self._out = dict(n=np.zeros(G, np.int32), idx=np.zeros((G, E.RES_CAP), np.int32), val=np.zeros((G, E.RES_CAP), np.float32),
                         best_idx=np.zeros(G, np.int32), action=np.zeros(G, np.int32))
        self._watch_net()
        self._set_tail()

    def _set_tail(self):
        """step_tail: the evaluate stage stops behind its logits / value_fc1 partial sums (model.forward_tail) and the step kernel
        finishes the row it consumes -- softmax and value, the bits bo_k_heads_rows would have written (Engine.step_heads): one
        launch and one pass over the [G, 4672] rows less per evaluation.  Engine softmax ("logits"), float32 evaluate stage with the
        hand-written head kernels, reference-semantics search; BETAONE_STEP_TAIL=0 keeps the stage's own last launch (A/B runs)."""
        m = self.model
        ok = (self.policy_kind == E.POLICY_LOGITS and not self.fast and not self.autocast and self.device.type == "cuda"
              and os.environ.get("BETAONE_STEP_TAIL", "1") != "0" and hasattr(m, "forward_tail") and m.tail_supported(self.G))
        self.step_tail = bool(ok)
        self._tail_params = tuple(m.tail_params()) if ok else None

    def _step_after(self, first, second):
        """The tree step behind an evaluation: (logits or probabilities, value) -> bo_step; (logits, partial sums) -> bo_step_heads."""
        if self.step_tail:
            self.eng.step_heads(first.data_ptr(), second.data_ptr(), *self._tail_params, second.shape[1], self.nn_in.data_ptr(), self._stream())
        else:
            self.eng.step(first.data_ptr(), second.data_ptr(), self.policy_kind, self.nn_in.data_ptr(), self._stream())

    def _watch_net(self):
        """The evaluate stage's own fault word (the split-precision tower: an activation beyond the fp16 range) travels with every
        result block the engine fetches (bo_engine_watch): checked once per ply by _check_watch, behind the copy the ply waits for
        anyway -- a net that saturates stops the run before a record made from its evaluations is handed out."""
        inner = getattr(self.model, "net", self.model)
        words = getattr(inner, "overflow_words", None)
        self._watch_msg = getattr(inner, "OVERFLOW_MESSAGE", "the evaluate stage reported a fault")
        ptr, n = words() if words is not None else (0, 1)
        self.eng.watch(ptr, n)

    def _check_watch(self):
        if self.eng.watch_seen():
            chk = getattr(getattr(self.model, "net", self.model), "check_overflow", None)
            if chk is not None and self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
                chk()  # (names the fault -- saturation, or which hand-off of the one-launch tower gave up -- and re-arms the stage's counters)
            raise E.EngineError(self._watch_msg)

    # ---- evaluate + step -------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _forward(self):
        if self.tower_timing is not None:  # (one host thread drives every cohort: the buffer named here is the one this forward's launch gets)
            getattr(self.model, "net", self.model).tower_timing_buf = self.tower_timing
        want_probs = self.policy_kind == E.POLICY_PROBS
        fused = getattr(self.model, "forward_probs", None) if (want_probs and not self.autocast) else None
        with torch.no_grad():
            if self.step_tail:  # (logits, value_fc1's partial sums [16, G, 256]): _step_after hands both to the step kernel
                logits, part = self.model.forward_tail(self.nn_in)
                return logits, part
            if fused is not None:  # the evaluate stage places the softmax itself (beside its value head)
                logits, value = fused(self.nn_in)
                want_probs = False
            elif self.autocast:
                with torch.autocast(self.device.type):
                    logits, value = self.model(self.nn_in)
            else:
                logits, value = self.model(self.nn_in)
        logits = logits.float()
        if want_probs:
            logits = torch.softmax(logits, dim=1)
        return logits.contiguous(), value.float().contiguous()

    def _eval_and_step_eager(self):
        self._stamp(1)
        logits, value = self._forward()
        self._stamp(2)
        self._logits, self._value = logits, value  # keep alive until the step kernel has run
        self._step_after(logits, value)
        self._stamp(3)

    def _capture(self):
        """Capture `net forward -> tree step` once; afterwards a step is one hipGraphLaunch."""
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside capture (MIOpen find, allocator)
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            logits, value = self._forward()
            self._step_after(logits, value)
        self._graph, self._logits, self._value = g, logits, value

    def _eval_and_step(self):
        self.n_forward += 1
        if self.use_graph:
            if self._graph is None:
                # the engine's state must not advance during capture: capture is recorded, not executed
                self._capture()
            self._graph.replay()
        else:
            self._eval_and_step_eager()

    MAX_GRAPH_ITERATIONS = 12

    def _eval_and_step_n(self, n: int):
        """n x (net forward -> tree step).  With graphs: ONE launch of a graph that holds the n iterations back to back (between
        two graph launches the device idles ~8 us -- profiles/r02_trace_percentiles.md -- between two nodes of one graph < 1 us)."""
        if n <= 1 or not self.use_graph:
            for _ in range(n):
                self._eval_and_step()
            return
        if n > self.MAX_GRAPH_ITERATIONS:  # (fast mode with few leaves per step: hundreds of evaluations per search)
            while n > 0:
                k = min(n, self.MAX_GRAPH_ITERATIONS)
                self._eval_and_step_n(k)
                n -= k
            return
        g = self._graphs_n.get(n)
        if g is None:
            if self._graph is None:
                self._capture()  # warms the allocator / MIOpen up as well
            cg, keep = torch.cuda.CUDAGraph(), []
            with torch.cuda.graph(cg, capture_error_mode=CAPTURE_MODE):
                for _ in range(n):
                    self._stamp(1)
                    logits, value = self._forward()
                    self._stamp(2)
                    self._step_after(logits, value)
                    self._stamp(3)
                    keep.append((logits, value))
            g = self._graphs_n[n] = (cg, keep)
        self.n_forward += n
        g[0].replay()

    # ---- game slots ---------------------------------------------------------------------------------
    @_on_main
    def start_games(self, slots: Sequence[int], game_ids: Sequence[int], rngs: Sequence, fens: Optional[Sequence] = None,
                    moves: Optional[Sequence] = None):
        self._start_games(slots, game_ids, rngs, fens, moves)

    def _start_games(self, slots, game_ids, rngs, fens=None, moves=None):  # (on torch's current stream: the refill path runs it on the side stream)
        self.eng.reset(list(slots), fens, moves, stream=self._stream())
        for i, s in enumerate(slots):
            fen = fens[i] if fens is not None else None
            full, white = _fen_meta(fen)
            self.games[s] = GameState(game_id=int(game_ids[i]), rng=rngs[i], start_fen=fen, start_fullmove=full,
                                      start_white=white)
            if moves is not None and moves[i]:
                self.games[s].plies = self.games[s].first_ply = len(moves[i].split())
            self._active[s] = True
            if self._begun_want is not None:
                self._begun_want[s] = False  # a search begun for the slot's previous occupant does not count
            self._plies[s] = self._first_ply[s] = self.games[s].plies
            self._start_full[s], self._start_black[s] = full, 0 if white else 1
            self._start_step[s] = self._step
            if self.rng_mode == "native":
                self.eng.rng_seed(s, int(rngs[i]))

    # ---- one search for every active slot (run_mcts) ----------------------------------------------------
    @_on_main
    def search(self, go: np.ndarray, n_legal: np.ndarray, terminal: np.ndarray) -> Dict[str, np.ndarray]:
        t0 = time.perf_counter()
        noise = None
        if self.alpha > 0:
            noise = np.zeros((self.G, E.MAX_LEGAL), dtype=np.float64)
            for g in np.nonzero(go)[0]:
                if terminal[g] == 0:  # mcts.py:179: noise only for a non-terminal root
                    noise[g, :n_legal[g]] = sampling.root_noise(n_legal[g], self.alpha, self.games[g].rng)
        self.host_seconds += time.perf_counter() - t0
        stream = self._stream()
        self.eng.search_begin(go, noise, self.nn_in.data_ptr(), stream)
        self.eng.step(0, 0, E.POLICY_NONE, self.nn_in.data_ptr(), stream)
        burst = self.expected_evals
        while True:
            for _ in range(burst):
                self._eval_and_step()
            running, _, _ = self.eng.poll(self._stream(), want_mask=False)
            if running == 0:
                break
            burst = 1
        self.n_sims += int(np.count_nonzero(go)) * self.S
        res = self.eng.result(self._stream())
        self._check_watch()
        return res

    # ---- one ply for every active game (the body of self_play.py:101-184) --------------------------------
    @_on_main
    def play_ply(self, on_finished: Optional[Callable[[FinishedGame], None]] = None,
                 refill: Optional[Callable[[int], Optional[tuple]]] = None,
                 while_searching: Optional[Callable[[], None]] = None) -> int:
        """Advance every active game by one move.  Finished games are reported through `on_finished`
        and their slots refilled by `refill(slot) -> (game_id, rng, fen)`.  `while_searching()` is the place for the caller's
        own per-ply host work (handing records on, logging): it runs once the ply's searches are enqueued and the finished
        games have been reported, while the device works -- code between two play_ply calls runs with the device idle.
        Returns the number of moves played."""
        if self.rng_mode == "native":
            return self._play_ply_native(on_finished, refill, while_searching)
        n_legal, terminal, ply = self.eng.root_info(self._stream())
        t0 = time.perf_counter()
        done_slots = [g for g in range(self.G) if self.games[g] is not None and
                      (terminal[g] != 0 or self.games[g].plies >= self.max_game_moves)]
        self.host_seconds += time.perf_counter() - t0
        if done_slots:
            new_slots, ids, rngs, fens = [], [], [], []
            for g in done_slots:
                fin = self._finish(g, int(terminal[g]))
                if on_finished is not None:
                    on_finished(fin)
                self.games[g] = None
                self._active[g] = False
                nxt = refill(g) if refill is not None else None
                if nxt is not None:
                    new_slots.append(g); ids.append(nxt[0]); rngs.append(nxt[1]); fens.append(nxt[2])
            if new_slots:
                self._start_games(new_slots, ids, rngs, fens)
                n_legal, terminal, ply = self.eng.root_info(self._stream())
        go = np.array([1 if (self.games[g] is not None and terminal[g] == 0) else 0 for g in range(self.G)], dtype=np.int32)
        if not go.any():
            if while_searching is not None:
                while_searching()
            return 0
        res = self.search(go, n_legal, terminal)
        t0 = time.perf_counter()
        actions = np.full(self.G, -1, dtype=np.int32)
        th, ti, tf = self.temperature
        for g in np.nonzero(go)[0]:
            gs = self.games[g]
            n = int(res["n"][g])
            idx, val = res["idx"][g, :n].copy(), res["val"][g, :n].copy()
            gs.pis.append((idx, val))
            actions[g] = sampling.select_action_sparse(idx, val, gs.fullmove_number(), gs.rng, th, ti, tf)
            gs.plies += 1
        self.host_seconds += time.perf_counter() - t0
        self.eng.play(actions, self._stream())
        if while_searching is not None:
            while_searching()
        n_moves = int(np.count_nonzero(go))
        self.n_plies += n_moves
        return n_moves

    def _forward_only(self):
        """The network forward alone (root evaluations of searches whose Dirichlet noise is still being drawn)."""
        if not self.use_graph:
            self._f_logits, self._f_value = self._forward()
            return
        if self._fgraph is None:
            if self._graph is None:
                self._capture()  # warms the allocator / MIOpen up as well
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                logits, value = self._forward()
            self._fgraph, self._f_logits, self._f_value = g, logits, value
        self._fgraph.replay()

    PLY_TRACE = []  # BO_PLY_PROFILE=trace: (time, cohort, phase-ended) of every phase boundary, every Rollout of the process

    def _pp(self, name):
        now = time.perf_counter()
        self.ply_profile.append((name, now - self._pp_t))
        if os.environ.get("BO_PLY_PROFILE") == "trace":
            Rollout.PLY_TRACE.append((self._pp_t, now, self._stamp_id, name))
        self._pp_t = now

    def _run_search_steps(self, poll: bool = True):
        """The evaluate -> step iterations of one search per game.  Returns the (n_legal, terminal, go) of searches that the
        previous turn began without waiting for the device (engine.LAZY_BEGIN), else None."""
        burst = self.expected_evals
        info = None
        if self._noise_pending:  # root evaluation: forward | host draws the noise meanwhile | upload | apply
            self._noise_pending = False
            if self._fwd_early:  # enqueued by the previous turn, before its host-side bookkeeping
                self._fwd_early = False
            else:
                self.n_forward += 1
                self._forward_only()
            if self._begun is E.LAZY_BEGIN:  # the roots' state, now that the device has its next 0.4 ms of work
                info = self.eng.selfplay_begun()
                self._begun = None
            if self.ply_profile is not None: self._pp("begun")
            self.eng.selfplay_noise(self._stream())
            if self.ply_profile is not None: self._pp("noise")
            self._step_after(self._f_logits, self._f_value)
            burst -= 1
        while True:
            self._eval_and_step_n(burst)
            if not poll:  # the caller asks the engine itself (bo_selfplay_turn, poll_first)
                return info
            running, _, _ = self.eng.poll(self._stream(), want_mask=False)
            if running == 0:
                break
            burst = 1

    def _play_ply_native(self, on_finished, refill, while_searching=None) -> int:
        """play_ply with the per-move host work done inside the library (same streams, same results)."""
        if self._turn_due is not None:  # (a ply begun through ply_begin and not ended yet)
            self.ply_end()
        if not self.ply_begin(on_finished, refill, while_searching):
            return 0
        return self.ply_end()

    @_on_main
    def ply_begin(self, on_finished=None, refill=None, while_searching=None) -> bool:
        """First half of a ply (native RNG mode): enqueue the ply's searches -- every evaluate -> step iteration the searches are
        expected to need -- hand finished games over and refill their slots beside the search, run the caller's `while_searching`,
        and RETURN WITHOUT WAITING for the device.  True: a turn is due (ply_end); False: no game searched this ply.
        play_ply() is ply_begin() + ply_end(); CohortRollout drives several Rollouts by ending one cohort's ply and beginning
        its next before it turns to the next cohort, so the device always holds enqueued work of the other cohorts."""
        assert self.rng_mode == "native" and self._turn_due is None
        eng, G = self.eng, self.G
        stream = self._stream()
        t0 = time.perf_counter()
        if self.ply_profile is not None:
            if self._t_ret is not None:
                self.ply_profile.append(("between_calls", t0 - self._t_ret))
            self._pp_t = t0
        want = self._active & (self._plies < self.max_game_moves)
        limit_done = np.nonzero(self._active & ~want)[0]
        self.host_seconds += time.perf_counter() - t0
        lazy = False
        if self._begun is not None:  # the previous turn already began these searches (bo_selfplay_turn)
            extra = want & ~self._begun_want
            if self._begun is E.LAZY_BEGIN and not extra.any():
                lazy = True  # ... without waiting for the device: their roots' state arrives with the first evaluation below
            else:
                if extra.any():
                    self._fwd_early = False  # (games started in between: their roots are not in the forward enqueued early -- it is repeated)
                nl, term, go = eng.selfplay_begun() if self._begun is E.LAZY_BEGIN else self._begun
                self._begun = None
                if extra.any():  # games started in between
                    nl2, t2, go2 = eng.selfplay_begin(extra.astype(np.int32), self.nn_in.data_ptr(), stream)
                    nl, term, go = np.where(extra, nl2, nl), np.where(extra, t2, term), go | go2
        else:
            nl, term, go = eng.selfplay_begin(want.astype(np.int32), self.nn_in.data_ptr(), stream)
        # Games that are over: their slots sit this ply out.  Exporting them, the caller's callback and setting up the next
        # games happen WHILE the GPU searches the other games' moves -- on a side stream: a finished slot's arrays are not
        # touched by the search (bo_k_step skips idle slots) -- and the new games' first searches are begun with everyone
        # else's next one by bo_selfplay_turn below.  (One idle slot-ply per game, ~0.3 % of the capacity, instead of an
        # idle GPU during ~0.3 ms of host work in every ply in which a game ends.)
        if lazy:
            nl, term, go = self._run_search_steps(poll=False)
        if self.ply_profile is not None: self._pp("enqueue_search")
        done = [int(g) for g in limit_done] + [int(g) for g in np.nonzero(want & (term != 0))[0]]
        for g in done:
            self._active[g] = False
        if not go.any():
            # (no search was begun: a root evaluation enqueued ahead for the begun searches evaluated nothing anyone will use, and no
            # Dirichlet draw is due -- a later ply must not take that forward for its own roots')
            self._fwd_early = False
            self._noise_pending = False
            self._finish_and_refill(done, term, on_finished, refill)
            if while_searching is not None:
                while_searching()
            return False
        if not lazy:
            self._run_search_steps(poll=False)
        if done:
            if self._side is not None:
                with torch.cuda.stream(self._side):  # torch's current stream in here: exports, encodes and set-up run beside the search
                    self._finish_and_refill(done, term, on_finished, refill)
            else:
                self._finish_and_refill(done, term, on_finished, refill)
        if self.ply_profile is not None: self._pp("finish_and_refill")
        if while_searching is not None:
            while_searching()
        if self.ply_profile is not None: self._pp("while_searching")
        self._turn_due = go
        if self._device_turn_ok():
            self._enqueue_autoturn(go)
        else:
            self._prefetch_result()
        self._mark_enqueued()
        if self.ply_profile is not None: self._pp("enqueue_turn")
        return True

    def _device_turn_ok(self) -> bool:
        return self.device_turn and self.eng.autoturn_supported(self.temperature)

    def _enqueue_autoturn(self, go: np.ndarray, redo: bool = False) -> None:
        """The ply's turn -- result, temperature sample, the played move, the next searches' begin (self_play.py:121-184, mcts.py:160-162) --
        enqueued on the DEVICE behind the searches, and behind it the next roots' evaluation: from a ply's last tree step to the next ply's first
        tower launch the device waits for nobody.  The host's share (the moves' uniforms now, the new roots' Dirichlet noise in the next
        ply_begin, while that evaluation runs) keeps every per-game RNG stream in the reference's order."""
        move_number = self._start_full + (self._plies + self._start_black) // 2   # board.fullmove_number, self_play.py:104
        want_next = self._active & ((self._plies + go) < self.max_game_moves)     # (finished games' slots have been refilled by now)
        self._auto_want = want_next
        self.eng.selfplay_autoturn(go, move_number, self.temperature, want_next.astype(np.int32), self.nn_in.data_ptr(), self._stream(), redo=redo)
        self._auto = True
        self._prefetched = False
        if want_next.any():  # the next searches' root evaluation (its noise is drawn and uploaded meanwhile, by the next ply_begin)
            self.n_forward += 1
            self._forward_only()
            self._fwd_early = True

    PREFETCH_RESULT = os.environ.get("BETAONE_RESULT_PREFETCH", "1") != "0"  # (0: the turn fetches the result block itself, for A/B runs)

    def _prefetch_result(self) -> None:
        """The result block's trip to pinned host memory, enqueued behind the searches' last expected step: by the time the stream is
        idle the turn finds it there (ply_end makes no device round trip of its own)."""
        self._prefetched = False
        if self.PREFETCH_RESULT and not self.fast:
            self.eng.result_prefetch(self._stream())
            self._prefetched = True

    def _mark_enqueued(self) -> None:
        if self.device.type == "cuda":
            if self._ply_event is None:
                self._ply_event = torch.cuda.Event()
            self._ply_event.record(torch.cuda.current_stream(self.device))

    def ply_ready(self) -> bool:
        """A ply is due and the device has finished everything ply_begin enqueued for it (never blocks): ply_end will not wait."""
        if self._turn_due is not None and self._auto:
            return self.eng.autoturn_ready()  # (the event right behind the turn's outputs, not behind the next root evaluation)
        return self._turn_due is not None and (self._ply_event is None or self._ply_event.query())

    @_on_main
    def ply_end(self, block: bool = True, drop: Sequence[int] = ()) -> Optional[int]:
        """Second half of a ply: wait for the searches ply_begin enqueued (the ply's ONE host round trip), sample and play the
        moves, begin the next searches on the device and enqueue their root evaluation.  Returns the number of moves played.
        block=False (CohortRollout's scheduler): if a search turns out to need one more evaluation than was enqueued, enqueue it
        and return None instead of waiting for it -- the ply stays due, ply_ready() tells when to call again.
        drop: slots whose game is being taken out of play (CohortRollout.retire): their search of the due ply counts for nothing --
        no pi, no ply -- so a retired game carries the records a single Rollout, which retires between two plies, would give it."""
        eng, G = self.eng, self.G
        stream = self._stream()
        go, self._turn_due = self._turn_due, None
        if len(drop):
            go = go.copy()
            go[list(drop)] = 0
        t0 = time.perf_counter()
        if self.ply_profile is not None: self._pp_t = t0
        move_number = self._start_full + (self._plies + self._start_black) // 2   # board.fullmove_number, self_play.py:104
        want_next = self._active & ((self._plies + go) < self.max_game_moves)     # the next call's `want`
        self.host_seconds += time.perf_counter() - t0
        # one native call: sample the moves, play them, begin the next searches (root info, Dirichlet noise, root planes)
        # (fast mode mixes the noise into a kept root's priors when the search begins, so its draws cannot be deferred)
        out = None
        if self._auto:  # the device made the turn itself: collect what it played and began (the ply's one wait, off the device's path)
            want_next = self._auto_want  # (as the enqueued turn was told)
            while True:
                out, begun = eng.autoturn_collect(self._out)
                if out is not None:
                    self._auto = False
                    break
                self._fwd_early = False
                if self.ply_profile is not None: self.ply_profile.append(("redo_turn", 0.0))
                self._check_watch()  # (the turn also holds still when the evaluate stage's fault word is set: raise, play nothing)
                self._eval_and_step()  # a search needed one more evaluation than expected: nothing was played; step, then the same turn again
                self._enqueue_autoturn(go, redo=True)
                if not block:
                    self._turn_due = go
                    self._mark_enqueued()
                    return None
        while out is None:  # one native call per ply: "all searches finished?" + sample + play + begin the next searches
            out, begun = eng.selfplay_turn(go, move_number, self.temperature, self._out, want_next.astype(np.int32), self.nn_in.data_ptr(), stream,
                                           defer_noise=not self.fast, poll_first=True, lazy_begin=not self.fast, prefetched=self._prefetched)
            self._prefetched = False
            if out is not None:
                break
            self._eval_and_step()  # a search needed one more evaluation than expected
            if not block:
                self._turn_due = go
                self._prefetch_result()
                self._mark_enqueued()
                return None
        self.n_sims += int(np.count_nonzero(go)) * self.S
        self._check_watch()  # (the evaluate stage's fault word arrived with the result block)
        if self.ply_profile is not None: self._pp("turn")
        t0 = time.perf_counter()
        actions = out["action"]
        if begun is None:  # rare: a pi not sparse enough for the native sampler -- nothing was played
            for g in np.nonzero(actions == -3)[0]:
                rs = np.random.RandomState(0)
                rs.set_state(eng.rng_get_state(int(g)))
                n = int(out["n"][g])
                th, ti, tf = self.temperature
                actions[g] = sampling.select_action_sparse(out["idx"][g, :n], out["val"][g, :n], int(move_number[g]), rs, th, ti, tf)
                eng.rng_set_state(int(g), rs.get_state())
            eng.play(actions, stream)
        else:
            self._begun, self._begun_want, self._noise_pending = begun, want_next.copy(), not self.fast
            if begun is E.LAZY_BEGIN and self._noise_pending and want_next.any():
                # the next searches' roots are on their way (begun without a host round trip): their evaluation follows at once, so the
                # device does not idle through this call's bookkeeping and the caller's time between two plies (~70 us per ply)
                self.n_forward += 1
                self._forward_only()
                self._fwd_early = True
                if self.ply_profile is not None: self._pp("early_forward")
        k = max(1, int(out["n"].max()))
        # (keyed by step WITH the mask of the games that searched: a slot refilled inside a ply sits that ply out, so its new
        # game's first pi belongs to the next step -- the row of this step is still its previous occupant's)
        self._hist[self._step] = (go.astype(bool), out["n"].copy(), out["idx"][:, :k].copy(), out["val"][:, :k].copy())
        if self._pk and k <= self._pk:  # this ply's pi of every searched game into its slot's row (one scatter)
            rows = np.nonzero(go)[0]
            cols = (self._plies[rows] - self._first_ply[rows]).astype(np.int64)
            if len(cols) and cols.max() >= self._pi_n.shape[1]:
                grow = max(2 * self._pi_n.shape[1], int(cols.max()) + 1)
                self._pi_n = np.concatenate([self._pi_n, np.zeros((G, grow - self._pi_n.shape[1]), np.int32)], axis=1)
                self._pi_idx = np.concatenate([self._pi_idx, np.zeros((G, grow - self._pi_idx.shape[1], self._pk), np.int32)], axis=1)
                self._pi_val = np.concatenate([self._pi_val, np.zeros((G, grow - self._pi_val.shape[1], self._pk), np.float32)], axis=1)
            self._pi_n[rows, cols] = out["n"][rows]
            self._pi_idx[rows, cols] = out["idx"][rows, :self._pk]
            self._pi_val[rows, cols] = out["val"][rows, :self._pk]
        elif self._pk:
            self._pk = 0  # a denser pi than the reference's search produces: fall back to the per-step history
        self._step += 1
        self._plies += go  # GameState.plies of the native mode is brought up to date when the game is finished
        lo = int(self._start_step[self._active].min()) if self._active.any() else self._step
        for st in [st for st in self._hist if st < lo]:
            del self._hist[st]
        self.host_seconds += time.perf_counter() - t0
        n_moves = int(np.count_nonzero(go))
        self.n_plies += n_moves
        if self.ply_profile is not None:
            self._pp("bookkeeping")
            self._t_ret = time.perf_counter()
        return n_moves

    def _finish_and_refill(self, done, term, on_finished, refill) -> None:
        new_slots, ids, seeds, fens = [], [], [], []
        for g in done:
            fin = self._finish(g, int(term[g]))
            if on_finished is not None:
                on_finished(fin)
            self.games[g] = None
            self._active[g] = False
            nxt = refill(g) if refill is not None else None
            if nxt is not None:
                new_slots.append(g); ids.append(nxt[0]); seeds.append(nxt[1]); fens.append(nxt[2])
        if new_slots:
            self._start_games(new_slots, ids, seeds, fens)  # (bo_games_reset synchronises its stream before it returns)

    def _finish(self, g: int, terminal: int) -> FinishedGame:
        gs = self.games[g]
        if self.rng_mode == "native":
            gs.plies = int(self._plies[g])
        positions, moves = self.eng.export_game(g, self._stream(), n_plies=gs.plies)
        outcome = 1.0 if terminal == 1 else 0.0
        pis = gs.pis
        if self.rng_mode == "native" and self._pk:  # this game's sparse pis: three slices of its slot's rows
            t = max(0, len(moves) - gs.first_ply)
            pis = SparsePis(self._pi_n[g, :t].copy(), self._pi_idx[g, :t].copy(), self._pi_val[g, :t].copy())
        elif self.rng_mode == "native":  # (fast mode: pi over many moves) gather from the per-step arrays
            pis = []
            for st in range(int(self._start_step[g]), self._step):
                went, n, idx, val = self._hist[st]
                if went[g]:
                    pis.append((idx[g, :n[g]].copy(), val[g, :n[g]].copy()))
        # one (state, pi) per move actually played (self_play.py:122,171): a start position that is already over yields
        # none, and a move the engine refused (no room left in the slot's position stack) leaves no record either
        pis = pis[:max(0, len(moves) - gs.first_ply)]
        return FinishedGame(game_id=gs.game_id, slot=g, moves=moves, positions=positions, pis=pis, outcome=outcome,
                            terminal=terminal, first_ply=gs.first_ply)

    # ---- training records ---------------------------------------------------------------------------------
    def encode_finished_in_slot(self, g: int, n_records: int, first_ply: int = 0) -> torch.Tensor:
        """Dense (state) tensors [n,120,8,8] of plies first_ply.. of the game still resident in slot g
        (self_play.py:200-208: END-of-game tracker).  Call from `on_finished`, i.e. before the slot is refilled."""
        out = torch.empty((max(1, n_records), E.INPUT_CHANNELS, 8, 8), dtype=torch.float32, device=self.device)
        if n_records:
            self.eng.encode_game(g, first_ply, n_records, out.data_ptr(), self._stream())
        return out[:n_records]

    @_on_main
    def retire(self, g: int, on_finished: Optional[Callable[[FinishedGame], None]] = None,
               refill: Optional[Callable[[int], Optional[tuple]]] = None) -> None:
        """Take the game in slot g out of play between two plies -- the reference's per-game abort paths
        (self_play.py:119,167,180 return None; :186 move limit) -- and hand the slot to the next game.  With `on_finished`
        the moves played so far are reported like a game stopped by the move limit; without it the game is dropped."""
        if self.games[g] is None:
            return
        if on_finished is not None:
            on_finished(self._finish(g, 0))
        self.games[g] = None
        self._active[g] = False
        nxt = refill(g) if refill is not None else None
        if nxt is not None:
            self.start_games([g], [nxt[0]], [nxt[1]], [nxt[2]])  # its first search is begun by the next play_ply

    @_on_main
    def swap_model(self, model: torch.nn.Module) -> None:
        """Replace the evaluate stage between two plies (weights handed over by the training side, main.py:147-148): every
        evaluation from the next play_ply on runs `model`; the captured graphs hold the old module's kernels and weight
        addresses, so they are dropped and re-captured on first use.  Searches already begun keep their trees."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)  # nothing of the old graphs is in flight when they are released
        self.model = model
        self._graph = self._fgraph = None
        self._graphs_n = {}
        self._logits = self._value = None
        self._fwd_early = False  # (a root evaluation enqueued early ran the old weights: it is repeated with the new ones)
        self._watch_net()
        self._set_tail()

    @_on_main
    def check_net(self):
        """Raise if the evaluate stage reports a fault of its own (the split-precision tower: an activation beyond the fp16 range).
        Reads the word on torch's current stream -- the one the forwards run on -- and waits for it: for the end of a run; during
        play the word is checked with every ply's result block (_check_watch)."""
        self._check_watch()
        chk = getattr(getattr(self.model, "net", self.model), "check_overflow", None)
        if chk is not None:
            chk()

    @_on_main
    def close(self):
        self._graph = self._fgraph = None
        self._graphs_n = {}
        try:
            self.check_net()
        finally:
            self.eng.close()


class _CohortEngines:
    """The `eng` of a CohortRollout: the per-slot queries callers make on Rollout.eng, spread over the cohorts' engines."""

    def __init__(self, parts):
        self._parts = parts

    @staticmethod
    def _beside(p) -> int:  # a stream that does not queue behind the cohort's enqueued searches (the status words are sticky ORs)
        return p._side.cuda_stream if p._side is not None else 0

    def status_bits(self, stream: int = 0) -> np.ndarray:
        return np.concatenate([p.eng.status_bits(self._beside(p)) for p in self._parts])

    def status(self, stream: int = 0) -> Dict[str, np.ndarray]:
        st = [p.eng.status(self._beside(p)) for p in self._parts]
        return {k: np.concatenate([s[k] for s in st]) for k in st[0]}

    @staticmethod
    def describe_status(bits: int) -> str:
        return E.Engine.describe_status(bits)

    def check_status(self) -> int:
        return sum(p.eng.check_status() for p in self._parts)


class CohortRollout:
    """G concurrent games as K cohorts of G / K slots, each cohort a Rollout of its own (engine, NN rows, captured graphs) on its
    own HIP stream, driven round-robin by one host thread with the plies software-pipelined: a call of play_ply ENDS each
    cohort's outstanding ply (the one host round trip: results, sampled moves, the next searches begun) and at once BEGINS its next
    one (every evaluate -> step iteration enqueued) before it turns to the next cohort.  While the host waits for cohort k and while
    cohort k runs its tree step, head kernels and ply boundary, the other cohorts' towers are on the device -- one cohort's serial
    chain tower -> heads -> step -> tower no longer leaves the matrix pipe dark.  Games are independent (the reference runs
    them in separate processes, main.py:160-175), so every game plays exactly the moves it plays in a single Rollout: same
    engine code, same per-game RNG streams; only WHEN a game's kernels run differs.

    Slot s belongs to cohort s // (G / K).  The interface is Rollout's (start_games / play_ply / retire / swap_model / close,
    games, n_sims, ...); play_ply ends one cohort-ply per cohort with work (in the order the device finishes them) and returns the moves played.
    `drain()` ends the outstanding plies without beginning new ones."""

    def __init__(self, model, n_games: int, cohorts: int = 2, cu_masks: Optional[str] = None, **kw):
        K = int(cohorts)
        if K < 1 or n_games % K:
            raise ValueError(f"CohortRollout: {n_games} games do not split into {K} equal cohorts")
        if kw.get("rng_mode", "native") != "native":
            raise ValueError("CohortRollout needs rng_mode='native' (the ply is split into ply_begin / ply_end)")
        kw["rng_mode"] = "native"
        self.device = E.runtime_device(kw.get("device", "cuda:0"))
        self.G, self.K, self.Gc = int(n_games), K, int(n_games) // K
        models = list(model) if isinstance(model, (list, tuple)) else [model] * K
        if K > 1 and len({id(m) for m in models}) < K and any(getattr(getattr(m, "net", m), "conv", None) == "tower_b1" for m in models):
            raise ValueError("CohortRollout: a conv='tower_b1' evaluate stage keeps its hand-off buffers in ONE handle (one launch in flight at "
                             "a time): pass one inference copy per cohort")
        # (cohort 0 keeps torch's current stream semantics only when it is alone; with K > 1 every cohort gets a stream of its own)
        # cu_masks: every cohort's stream confined to its own 1/K of the compute units ("contiguous" / "interleaved", engine.
        # cu_partition_masks); None / "off": plain streams, the dispatcher places the cohorts' workgroups as it likes.
        # "auto" (default, or BETAONE_COHORT_CU_MASK): contiguous from three cohorts up.  What matters most is that a stream made with a CU
        # mask has a hardware queue of its own (torch's pool streams share queues: four cohorts then wait for each other's launches, 4.1-4.7
        # ms per ply against 3.0); the confinement itself is worth <= 0.7 % -- profiles/r04_cohort_cu_masks.md.  "full": every CU for every cohort.
        self.cu_masks = cu_masks if cu_masks is not None else os.environ.get("BETAONE_COHORT_CU_MASK", "auto")
        if self.cu_masks == "auto":
            self.cu_masks = "contiguous" if K > 2 else "off"
        if self.cu_masks not in ("off", "contiguous", "interleaved", "full"):
            raise ValueError(f"CohortRollout: cu_masks={self.cu_masks!r} (off / contiguous / interleaved / auto)")
        self._masked: List[E.MaskedStream] = []
        if K > 1 and self.device.type == "cuda" and self.cu_masks != "off":
            n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
            try:
                self._masked = [E.MaskedStream(self.device, m) for m in E.cu_partition_masks(n_cu, K, self.cu_masks)]
            except E.EngineError as ex:  # (placement only: the games come out the same on pool streams, slower from three cohorts up)
                import warnings
                for m in self._masked:
                    m.close()
                self._masked = []
                warnings.warn(f"CohortRollout: no CU-masked streams ({ex}); the cohorts run on torch's pool streams", RuntimeWarning)
                self.cu_masks = "off"
        self.parts: List[Rollout] = []
        for k in range(K):
            st = None
            if K > 1 and self.device.type == "cuda":
                st = self._masked[k].stream if self._masked else torch.cuda.Stream(self.device)
            self.parts.append(Rollout(models[k], self.Gc, stream=st, **kw))
        if K > 1 and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)  # (buffers zeroed on the constructing stream are used on the cohorts' streams from here on)
        self.eng = _CohortEngines(self.parts)
        self._rr = 0
        self.wait_seconds = 0.0
        p0 = self.parts[0]
        self.S, self.B, self.L, self.fast = p0.S, p0.B, p0.L, p0.fast
        self.expected_evals, self.rng_mode, self.max_game_moves = p0.expected_evals, p0.rng_mode, p0.max_game_moves

    # ---- Rollout's counters and per-slot views --------------------------------------------------------------------------------
    n_sims = property(lambda self: sum(p.n_sims for p in self.parts))
    n_plies = property(lambda self: sum(p.n_plies for p in self.parts))
    n_forward = property(lambda self: sum(p.n_forward for p in self.parts))
    host_seconds = property(lambda self: sum(p.host_seconds for p in self.parts))
    games = property(lambda self: [g for p in self.parts for g in p.games])

    @property
    def ply_profile(self):  # BO_PLY_PROFILE=1: the cohorts' host phases, one list
        if self.parts[0].ply_profile is None:
            return None
        return [x for p in self.parts for x in p.ply_profile]

    @property
    def use_graph(self):
        return self.parts[0].use_graph

    @use_graph.setter
    def use_graph(self, v):
        for p in self.parts:
            p.use_graph = v

    def _split(self, slots):
        """global slots -> {cohort: [positions in `slots`]}"""
        by = {}
        for i, s in enumerate(slots):
            by.setdefault(int(s) // self.Gc, []).append(i)
        return by

    def start_games(self, slots, game_ids, rngs, fens=None, moves=None):
        for k, pos in self._split(slots).items():
            pick = lambda seq: [seq[i] for i in pos] if seq is not None else None
            self.parts[k].start_games([int(slots[i]) - k * self.Gc for i in pos], pick(game_ids), pick(rngs), pick(fens), pick(moves))

    def _callbacks(self, k, on_finished, refill):
        base = k * self.Gc

        def fin_cb(fin):
            fin.slot += base  # (the slot the CALLER knows; encode_finished_in_slot maps it back)
            on_finished(fin)

        return (fin_cb if on_finished is not None else None), ((lambda s: refill(s + base)) if refill is not None else None)

    def play_ply(self, on_finished=None, refill=None, while_searching=None) -> int:
        """One cohort-ply per cohort with work, ended in the order in which the DEVICE finishes them: a cohort whose ply is done is
        turned and begun again at once, whichever it is (in a fixed rotation the host sat waiting for cohort 0 while cohort 2 had
        finished); a search that needs one more evaluation than was enqueued gets it without the host waiting for it (ply_end(block=
        False)).  Every cohort is turned exactly once per call (letting fast cohorts run plies ahead of slow ones measured the same,
        profiles/r04_cohort_cu_masks.md).  The first call begins every cohort's ply before it ends any."""
        moved = 0
        cbs = [self._callbacks(k, on_finished, refill) for k in range(self.K)]
        for k, p in enumerate(self.parts):
            if p._turn_due is None:
                p.ply_begin(*cbs[k])
        pending = [k for k, p in enumerate(self.parts) if p._turn_due is not None]
        if while_searching is not None:
            while_searching()
        while pending:
            k = self._pick_ready(pending)
            p = self.parts[k]
            n = p.ply_end(block=False)
            if n is None:  # (one more evaluation was enqueued for a straggling search: the cohort stays due)
                continue
            moved += n
            pending.remove(k)
            p.ply_begin(*cbs[k])
        return moved

    def _pick_ready(self, cands) -> int:
        """The first of `cands` (rotating start) whose enqueued work the device has finished; polls without blocking, then yields."""
        t0 = None
        while True:
            for i in range(len(cands)):
                k = cands[(self._rr + i) % len(cands)]
                if self.parts[k].ply_ready():
                    self._rr = (self._rr + i + 1) % max(1, len(cands))
                    if t0 is not None:
                        self.wait_seconds += time.perf_counter() - t0  # (the host's time without a ready cohort: the device is the bottleneck then)
                    return k
            # Cohorts come due every ~0.7 ms: poll on (yielding the core between polls -- a timer sleep's wake-up latency would land on the
            # device's critical path) for 2 ms; only a wait longer than that (a long search, a stalled device) backs off to 50 us sleeps,
            # so that an idle rank does not hold a core at 100 % for the side threads' and the other ranks' account.
            now = time.perf_counter()
            if t0 is None:
                t0 = now
            time.sleep(0 if now - t0 < 2e-3 else 50e-6)

    def drain(self) -> int:
        """End every outstanding ply (no new searches are enqueued): the state then is what a sequence of whole plies leaves."""
        return sum(p.ply_end() for p in self.parts if p._turn_due is not None)

    def encode_finished_in_slot(self, g: int, n_records: int, first_ply: int = 0):
        return self.parts[g // self.Gc].encode_finished_in_slot(g % self.Gc, n_records, first_ply)

    def retire(self, g: int, on_finished=None, refill=None) -> None:
        k = g // self.Gc
        fin_cb, refill_cb = self._callbacks(k, on_finished, refill)
        if self.parts[k]._turn_due is not None:
            # a game leaves between two plies: its cohort's outstanding ply is ended first -- without a turn for the leaving game (the
            # single Rollout retires it before that ply begins: same moves, same records; a turn the device has already made for the
            # slot was refused there for the reason the game is retired, or is discarded with the slot's reset)
            self.parts[k].ply_end(drop=[g % self.Gc])
        self.parts[k].retire(g % self.Gc, fin_cb, refill_cb)

    def swap_model(self, model) -> None:
        """Rollout.swap_model for every cohort.  Every cohort has a whole ply enqueued on its own stream (graph replays that hold the old
        module's kernels and weight addresses, the next root evaluation): those plies are ENDED first -- played with the old weights, like a
        single Rollout's last ply before the swap -- so nothing of the old graphs is in flight when they are released; every search begun
        after this call evaluates with `model`."""
        models = list(model) if isinstance(model, (list, tuple)) else [model] * self.K
        self.drain()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for p, m in zip(self.parts, models):
            p.swap_model(m)

    def check_net(self):
        for p in self.parts:
            p.check_net()

    def close(self):
        err = None
        for p in self.parts:
            try:
                p.close()
            except Exception as ex:  # (close every engine; report the first fault)
                err = err or ex
        for m in self._masked:  # (graphs captured on these streams went with the parts above)
            m.close()
        self._masked = []
        if err is not None:
            raise err
