"""
betaone_amd/records.py -- finished games as compact records, and their exchange between GPUs.

The reference hands self-play results to training through the file system: one pickle per game of
dense `(state f32[120,8,8], pi f32[4672], z)` tuples, 49.4 KB per ply (self_play.py:220-231 ->
train.py:187-219).  On an 8-GPU node the only exchange step of the path is the same hand-over, done
as ONE RCCL all-gather of compact records over xGMI: a ply is its position
(88 B of bitboards and counters) + the sparse pi (<= 2 entries with the reference's search) + z; the
dense planes are a pure function of the game's position list and are re-expanded on the receiving
GPU by the engine's encode kernel (bo_records_encode).  ~100 B/ply instead of 49 KB/ply, so the ring
all-gather (per-link bound, ~153 GB/s on xGMI) moves megabytes, not gigabytes.

Wire format of one game (little endian):
  int32 magic 'BOG1' | int32 game_id | int32 n_plies | int32 terminal | float32 outcome | int32 n_pi_entries
  bo_position positions[n_plies + 1]                      (88 B each)
  int32 moves[n_plies]
  int32 pi_ptr[n_plies + 1] | int32 pi_idx[n_pi_entries] | float32 pi_val[n_pi_entries]
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import engine as E
from . import sampling

MAGIC = 0x31474F42  # 'BOG1'
POS_BYTES = C.sizeof(E.BoPosition)


def pack_game(fin) -> bytes:
    """FinishedGame (betaone_amd.rollout) -> bytes."""
    if getattr(fin, "first_ply", 0):
        raise ValueError("pack_game: a game continued from a move prefix has no pi for its first plies")
    n = len(fin.pis)
    if hasattr(fin.pis, "flat"):  # rollout.SparsePis: already arrays
        ptr, idx, val = fin.pis.flat()
    else:
        ptr = np.zeros(n + 1, dtype=np.int32)
        for i, (ix, _) in enumerate(fin.pis):
            ptr[i + 1] = ptr[i] + len(ix)
        idx = np.concatenate([np.asarray(i, dtype=np.int32) for i, _ in fin.pis]) if n else np.zeros(0, np.int32)
        val = np.concatenate([np.asarray(v, dtype=np.float32) for _, v in fin.pis]) if n else np.zeros(0, np.float32)
    head = np.array([MAGIC, fin.game_id, n, fin.terminal, 0, len(idx)], dtype=np.int32)
    head[4:5].view(np.float32)[0] = fin.outcome
    raw = getattr(fin.positions, "raw", None)  # engine.PositionList: one ctypes array, no per-position objects
    pos = bytes(raw)[:(n + 1) * POS_BYTES] if raw is not None else b"".join(bytes(p) for p in fin.positions[:n + 1])
    return b"".join([head.tobytes(), pos, np.asarray(fin.moves[:n], dtype=np.int32).tobytes(), ptr.tobytes(),
                     idx.tobytes(), val.tobytes()])


def unpack_games(buf: bytes) -> List[dict]:
    out, off = [], 0
    mv = memoryview(buf)
    while off + 24 <= len(buf):
        head = np.frombuffer(mv[off:off + 24], dtype=np.int32)
        if head[0] != MAGIC:
            break
        gid, n, term, nent = int(head[1]), int(head[2]), int(head[3]), int(head[5])
        outcome = float(head[4:5].view(np.float32)[0])
        off += 24
        positions = (E.BoPosition * (n + 1)).from_buffer_copy(mv[off:off + POS_BYTES * (n + 1)])
        off += POS_BYTES * (n + 1)
        moves = np.frombuffer(mv[off:off + 4 * n], dtype=np.int32).copy(); off += 4 * n
        ptr = np.frombuffer(mv[off:off + 4 * (n + 1)], dtype=np.int32).copy(); off += 4 * (n + 1)
        idx = np.frombuffer(mv[off:off + 4 * nent], dtype=np.int32).copy(); off += 4 * nent
        val = np.frombuffer(mv[off:off + 4 * nent], dtype=np.float32).copy(); off += 4 * nent
        out.append(dict(game_id=gid, n_plies=n, terminal=term, outcome=outcome, positions=positions, moves=moves,
                        pis=[(idx[ptr[i]:ptr[i + 1]], val[ptr[i]:ptr[i + 1]]) for i in range(n)]))
    return out


def expand_game(game: dict, device="cuda:0") -> List[Tuple[torch.Tensor, np.ndarray, float]]:
    """Compact record -> the reference's dense SelfPlayData list (self_play.py:200-216), planes by the
    engine's encode kernel on `device`."""
    lib = E.load_hip_library()
    n = game["n_plies"]
    if n == 0:
        return []
    dev = torch.device(device)
    out = torch.empty((n, E.INPUT_CHANNELS, 8, 8), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    rc = lib.bo_records_encode(n + 1, game["positions"], 0, n, out.data_ptr(), stream)
    if rc != 0:
        raise E.EngineError(f"bo_records_encode: {lib.bo_last_error().decode()}")
    states = out.cpu()
    recs = []
    for i in range(n):
        idx, val = game["pis"][i]
        z = game["outcome"] if game["positions"][i].turn == 1 else -game["outcome"]
        recs.append((states[i].clone(), sampling.dense_pi(idx, val), z))
    return recs


# ---- on disk ---------------------------------------------------------------------------------
# The reference writes one pickle per game of DENSE tuples, 49.4 KB per ply (self_play.py:220-231), and train.py reads them all
# back into one Python list (train.py:187-219).  The compact form on disk is the wire format above, games simply concatenated
# (a file can be appended to and two files can be cat'ed together): ~100 B per ply.  CompactDataset yields exactly the triple
# ChessDataset.__getitem__ yields (train.py:179-184), planes re-expanded by the engine's encode kernel one game at a time.
COMPACT_SUFFIX = ".bog"


def compact_path(data_dir: str, iteration: int, rank: int = 0) -> str:
    """DATA_DIR/iter_{iteration}/games_rank{rank}.bog -- beside the reference's game_{id}.pkl files (self_play.py:224-226)."""
    import os

    return os.path.join(data_dir, f"iter_{iteration}", f"games_rank{rank}{COMPACT_SUFFIX}")


_KNOWN_COMPLETE: Dict[str, int] = {}  # path -> its size after this process's last complete append


def complete_prefix_bytes(path: str) -> int:
    """Length of the leading run of complete games in the compact file `path` (0 for a missing / empty file): where the next
    game has to be appended.  A writer killed inside a write leaves a partial record behind the last complete game."""
    import os

    if not os.path.exists(path):
        return 0
    with open(path, "rb") as fh:
        idx = scan_games(fh.read())
    return idx[-1][2] + idx[-1][3] if idx else 0


def save_games(path: str, finished: Sequence, append: bool = True) -> int:
    """Append the compact records of `finished` (rollout.FinishedGame objects, or already packed bytes) to `path`; returns the
    bytes written.  One write per call, flushed: a reader (or a resume) never sees half a game from a finished call.  A
    partial record left at the tail by a writer that was killed inside its write is cut off first (the file is truncated to
    its last complete game): bytes appended behind it would be read as the rest of that record, and every later game lost."""
    import os

    blob = b"".join(f if isinstance(f, (bytes, bytearray)) else pack_game(f) for f in finished)
    if not blob:
        return 0
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    key = os.path.abspath(path)
    size = os.path.getsize(path) if os.path.exists(path) else 0
    # (a file this process left complete is not scanned again: the scan is for what a PREVIOUS, killed writer may have left behind)
    keep = 0 if not append else (size if _KNOWN_COMPLETE.get(key) == size else complete_prefix_bytes(path))
    with open(path, "r+b" if (append and os.path.exists(path)) else "wb") as fh:
        fh.truncate(keep)
        fh.seek(keep)
        fh.write(blob)
        fh.flush()
        os.fsync(fh.fileno())
    _KNOWN_COMPLETE[key] = keep + len(blob)
    return len(blob)


def scan_games(buf) -> List[Tuple[int, int, int, int]]:
    """(game_id, n_plies, byte offset, byte length) of every complete game in a compact buffer -- headers only, nothing unpacked
    (a truncated tail, e.g. from a killed writer, ends the scan)."""
    out, off, mv = [], 0, memoryview(buf)
    while off + 24 <= len(mv):
        head = np.frombuffer(mv[off:off + 24], dtype=np.int32)
        if head[0] != MAGIC:
            break
        n, nent = int(head[2]), int(head[5])
        size = 24 + POS_BYTES * (n + 1) + 4 * n + 4 * (n + 1) + 8 * nent
        if n < 0 or nent < 0 or off + size > len(mv):
            break
        out.append((int(head[1]), n, off, size))
        off += size
    return out


def load_games(path: str) -> List[dict]:
    """Every complete game of a compact file (a truncated tail is ignored, as by scan_games)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    idx = scan_games(buf)
    return unpack_games(buf[:idx[-1][2] + idx[-1][3]]) if idx else []


def game_ids_on_disk(data_dir: str, iteration: int) -> set:
    """Ids of the games of `iteration` that already have a compact record in any games_rank*.bog (resume, main.py:26-36)."""
    import glob
    import os

    ids = set()
    for p in glob.glob(os.path.join(data_dir, f"iter_{iteration}", f"games_rank*{COMPACT_SUFFIX}")):
        with open(p, "rb") as fh:
            ids.update(g[0] for g in scan_games(fh.read()))
    return ids


class CompactDataset(torch.utils.data.Dataset):
    """torch Dataset over compact game files: item i = the i-th ply of the concatenated games, as the triple
    ChessDataset.__getitem__ returns (train.py:179-184): (state float32 [120,8,8], policy float32 tensor [4672], value float32
    tensor [1]) -- bit-identical to what the reference's pickle of the same game yields.  A game's planes are re-expanded on
    `device` by the engine's encode kernel (bo_records_encode: history blocks, END-of-game repetition counts, self_play.py:200-208)
    the first time one of its plies is asked for, and the last `cache_games` expanded games are kept (a DataLoader that walks
    the plies in order, or shuffles within a window of games, expands each game once)."""

    def __init__(self, paths: Sequence[str], device="cuda:0", cache_games: int = 64):
        self.device, self.cache_games = device, max(1, int(cache_games))
        self._blobs, self._index = [], []  # file contents; per game (blob, offset, length, plies)
        for p in ([paths] if isinstance(paths, str) else list(paths)):
            with open(p, "rb") as fh:
                b = fh.read()
            self._blobs.append(b)
            self._index.extend((len(self._blobs) - 1, off, size, n) for _gid, n, off, size in scan_games(b) if n > 0)
        self._first = np.zeros(len(self._index) + 1, dtype=np.int64)  # first item of each game
        np.cumsum([g[3] for g in self._index], out=self._first[1:])
        self._cache = {}  # game number -> dense records (insertion-ordered: oldest first)

    def __len__(self) -> int:
        return int(self._first[-1])

    def game_of(self, i: int) -> Tuple[int, int]:
        g = int(np.searchsorted(self._first, i, side="right")) - 1
        return g, i - int(self._first[g])

    def _expand(self, g: int):
        recs = self._cache.get(g)
        if recs is None:
            blob, off, size, _n = self._index[g]
            recs = expand_game(unpack_games(self._blobs[blob][off:off + size])[0], self.device)
            self._cache[g] = recs
            while len(self._cache) > self.cache_games:
                self._cache.pop(next(iter(self._cache)))
        return recs

    def __getitem__(self, i: int):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        g, k = self.game_of(i)
        state, policy, value = self._expand(g)[k]
        return state, torch.from_numpy(policy).float(), torch.tensor([value], dtype=torch.float32)  # train.py:181-184


class GpuReplayBuffer:
    """Finished games resident in HBM as compact records, training batches expanded on the device (csrc/bo_replay.h, SURVEY.md
    section 8f row f3): what train.load_recent_data + ChessDataset + DataLoader do on the host with 49.4 KB per ply
    (/root/reference/train.py:179-219) done where the batches are consumed, with ~110 B per ply.

        buf = GpuReplayBuffer(capacity_plies=2_000_000)
        buf.add(games)                                  # rollout.FinishedGame objects, or the dicts unpack_games / load_games return
        for states, policies, values in buf.loader(batch_size=256, steps=1000, seed=0):   # train_network's loop, train.py:252-262
            ...                                         # float32 CUDA tensors [B,120,8,8], [B,4672], [B,1] -- ChessDataset's triple, batched

    A batch item is bit-identical to what the reference's pickle of the same game yields through ChessDataset.__getitem__ (planes with
    the END-of-game repetition counts, dense pi, z with its sign).  The oldest games leave when the buffer is full (the reference keeps
    the most recent iterations, train.py:190-193).  pi_width: most entries a pi may have (2 with the reference's search)."""

    def __init__(self, capacity_plies: int, device="cuda:0", pi_width: int = 2):
        self.lib = E.load_hip_library()
        self.device = E.runtime_device(device)
        self.pi_width = int(pi_width)
        h = C.c_void_p()
        # (a game of n records takes n + 1 position slots: room for capacity_plies records of games of ~64 plies and longer)
        slots = int(capacity_plies) + max(2, int(capacity_plies) // 64)
        idx = self.device.index if self.device.index is not None else 0
        self._check(self.lib.bo_replay_create(slots, self.pi_width, idx, C.byref(h)))
        self.h = h
        self.n_evicted = 0

    def _check(self, rc: int):
        if rc != 0:
            raise E.EngineError(f"replay buffer: {self.lib.bo_last_error().decode()}")

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.bo_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n, g = C.c_int64(), C.c_int64()
        self._check(self.lib.bo_replay_size(self.h, C.byref(n), C.byref(g)))
        return int(n.value)

    @property
    def n_games(self) -> int:
        n, g = C.c_int64(), C.c_int64()
        self._check(self.lib.bo_replay_size(self.h, C.byref(n), C.byref(g)))
        return int(g.value)

    def add(self, games: Sequence) -> int:
        """Add finished games (FinishedGame objects, packed bytes, or unpack_games dicts); returns the records evicted to make room."""
        lost = 0
        for g in games:
            if isinstance(g, (bytes, bytearray)):
                lost += self.add(unpack_games(bytes(g)))
                continue
            if not isinstance(g, dict):
                g = unpack_games(pack_game(g))[0]
            n = int(g["n_plies"])
            if n == 0:
                continue
            ptr = np.zeros(n + 1, dtype=np.int32)
            np.cumsum([len(ix) for ix, _ in g["pis"]], out=ptr[1:])
            idx = np.concatenate([np.asarray(ix, dtype=np.int32) for ix, _ in g["pis"]]) if ptr[-1] else np.zeros(1, np.int32)
            val = np.concatenate([np.asarray(v, dtype=np.float32) for _, v in g["pis"]]) if ptr[-1] else np.zeros(1, np.float32)
            out = np.float32(g["outcome"])
            z = np.array([out if g["positions"][i].turn == 1 else -out for i in range(n)], dtype=np.float32)  # self_play.py:202
            ev = C.c_int64(0)
            self._check(self.lib.bo_replay_add_game(self.h, int(g["game_id"]), g["positions"], n, ptr.ctypes.data_as(E._I32P),
                                                    np.ascontiguousarray(idx).ctypes.data_as(E._I32P), np.ascontiguousarray(val).ctypes.data_as(E._F32P),
                                                    z.ctypes.data_as(E._F32P), C.byref(ev), self._stream()))
            lost += int(ev.value)
        self.n_evicted += lost
        return lost

    def batch(self, record_index) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(states [n,120,8,8], policies [n,4672], values [n,1]) of the records with these indices (0 = oldest resident record)."""
        q = np.ascontiguousarray(record_index, dtype=np.int64).reshape(-1)
        n = int(q.size)
        kw = dict(dtype=torch.float32, device=self.device)
        states, pis, zs = torch.empty((n, E.INPUT_CHANNELS, 8, 8), **kw), torch.empty((n, E.NUM_ACTIONS), **kw), torch.empty((n, 1), **kw)
        self._check(self.lib.bo_replay_sample(self.h, n, q.ctypes.data_as(C.POINTER(C.c_int64)), states.data_ptr(), pis.data_ptr(), zs.data_ptr(),
                                              self._stream()))
        return states, pis, zs

    def sample(self, batch_size: int, rng: Optional[np.random.Generator] = None):
        rng = rng if rng is not None else np.random.default_rng()
        return self.batch(rng.integers(0, len(self), size=int(batch_size)))

    def loader(self, batch_size: int, steps: Optional[int] = None, seed: Optional[int] = None, shuffle: bool = True):
        """An iterable with DataLoader's contract for train_network (train.py:252: `for states, t_policies, t_values in dataloader`):
        one epoch over the resident records in a random order (shuffle=True, the reference's DataLoader(shuffle=True)), or `steps` batches
        drawn with replacement.  Batches are made on the buffer's device; the loop's `.to(config.DEVICE)` finds them there."""
        return _ReplayLoader(self, int(batch_size), steps, seed, shuffle)


class _ReplayLoader:
    def __init__(self, buf: GpuReplayBuffer, batch_size: int, steps, seed, shuffle):
        self.buf, self.batch_size, self.steps, self.seed, self.shuffle = buf, batch_size, steps, seed, shuffle

    def __len__(self) -> int:
        return self.steps if self.steps is not None else (len(self.buf) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        rng = np.random.default_rng(self.seed)
        n = len(self.buf)
        if self.steps is not None:
            for _ in range(self.steps):
                yield self.buf.batch(rng.integers(0, n, size=self.batch_size))
            return
        order = rng.permutation(n) if self.shuffle else np.arange(n)
        for i in range(0, n, self.batch_size):
            yield self.buf.batch(order[i:i + self.batch_size])


def all_gather_bytes(payload: bytes, device: Optional[torch.device] = None, group=None) -> List[bytes]:
    """One exchange step: every rank contributes `payload`, every rank receives all payloads.
    Two collectives: sizes (int64 all_gather), then the padded payload (uint8 all_gather) -- RCCL over
    xGMI when the process group's backend is nccl, gloo on CPU in the tests."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return [payload]
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = device if (device is not None and backend == "nccl") else torch.device("cpu")
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes_h = sizes.cpu().tolist()
    mx = max(sizes_h)
    if mx == 0:
        return [b""] * world
    pad = torch.zeros(mx, dtype=torch.uint8)
    if payload:
        pad[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    pad = pad.to(dev)
    gathered = torch.empty(world * mx, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    g = gathered.cpu().numpy()
    return [g[r * mx:r * mx + sizes_h[r]].tobytes() for r in range(world)]


def all_gather_games(finished: Sequence, device: Optional[torch.device] = None, group=None) -> List[dict]:
    """All ranks' finished games of this flush, as compact records, on every rank."""
    blobs = all_gather_bytes(b"".join(pack_game(f) for f in finished), device, group)
    games: List[dict] = []
    for b in blobs:
        games.extend(unpack_games(b))
    return games


class ExchangeError(RuntimeError):
    """The record exchange cannot complete (a peer rank died or hangs): the caller must stop -- exit non-zero -- not wait."""


def exchange_timeout_s() -> float:
    import os

    return float(os.environ.get("BETAONE_EXCHANGE_TIMEOUT", "120"))


class _Gather:
    """One all-gather in flight: `src` (uint8 / int64, equal length on every rank) -> host array, without a blocking call
    on the issuing side.  RCCL: H2D, collective and D2H all run on a side stream behind an event, through pinned and
    device staging buffers that the owner keeps and re-uses (`bufs`, grow-only: pinned allocations are slow); gloo: async
    work handle."""

    def __init__(self, src: np.ndarray, world: int, group, device: Optional[torch.device], side, bufs: Optional[dict] = None):
        import torch.distributed as dist

        self.n, self.world = src.size, world
        tdt = torch.from_numpy(src[:0]).dtype
        if device is not None:  # backend nccl (= RCCL over xGMI)
            key = str(tdt)
            b = bufs.get(key) if bufs is not None else None
            if b is None or b[0].numel() < src.size:
                cap = max(64, 2 * src.size)
                b = (torch.empty(cap, dtype=tdt).pin_memory(), torch.empty(cap, dtype=tdt, device=device),
                     torch.empty(world * cap, dtype=tdt, device=device), torch.empty(world * cap, dtype=tdt).pin_memory())
                if bufs is not None:
                    bufs[key] = b
                side.wait_stream(torch.cuda.current_stream(device))  # (new device buffers come from the compute stream's allocator)
            self.h_in, self.d_in = b[0][:src.size], b[1][:src.size]
            self.d_out, self.h_out = b[2][:world * src.size], b[3][:world * src.size]
            self.h_in.numpy()[:] = src
            self.event = torch.cuda.Event()
            # The side stream does NOT wait for the compute stream: nothing here depends on it, and behind a whole ply of queued
            # search the staging copies and the collective would start exactly at the ply boundary, in front of the copies of the
            # one host round trip the ply has (measured: +0.27 ms per ply on every rank).
            with torch.cuda.stream(side):
                self.d_in.copy_(self.h_in, non_blocking=True)
                work = dist.all_gather_into_tensor(self.d_out, self.d_in, group=group, async_op=True)
                work.wait()  # the SIDE stream waits for the collective; the host and the compute stream do not
                self.h_out.copy_(self.d_out, non_blocking=True)
                self.event.record(side)
            self.work = None
        else:
            self.h_in = torch.from_numpy(src)
            self.h_out = torch.empty(world * src.size, dtype=self.h_in.dtype)
            self.work = dist.all_gather_into_tensor(self.h_out, self.h_in, group=group, async_op=True)
            self.event = None

    def done(self) -> bool:
        return self.event.query() if self.event is not None else self.work.is_completed()

    def result(self, timeout_s: Optional[float] = None) -> np.ndarray:
        """[world, n] view of the staging buffer: consume (copy out) before the next gather of the same dtype starts.  A
        collective that is still unfinished after `timeout_s` (a peer rank died or hangs mid-period) raises ExchangeError
        instead of waiting for ever: RCCL path = the completion event is polled; gloo = the work handle's own timeout."""
        import datetime
        import time as _time

        timeout_s = exchange_timeout_s() if timeout_s is None else float(timeout_s)
        if self.event is not None:
            deadline = _time.monotonic() + timeout_s  # already complete when a whole exchange period of plies ran in between
            while not self.event.query():
                if _time.monotonic() > deadline:
                    raise ExchangeError(f"record all-gather still unfinished after {timeout_s:.0f} s: a peer rank died or hangs")
                _time.sleep(0.0005)
        else:
            try:
                ok = self.work.wait(datetime.timedelta(seconds=timeout_s))
            except Exception as ex:  # gloo reports a dead peer ("Connection closed by peer") or its timeout as an exception
                raise ExchangeError(f"record all-gather failed: {type(ex).__name__}: {str(ex).splitlines()[0] if str(ex) else ''}") from ex
            if ok is False:
                raise ExchangeError(f"record all-gather still unfinished after {timeout_s:.0f} s: a peer rank died or hangs")
        return self.h_out.numpy().reshape(self.world, self.n)


class PeriodicGameExchange:
    """The path's only exchange step -- finished games' compact records to every rank -- taken off
    the critical path of the plies.  Every rank calls `push(finished)` once per ply; records accumulate on the host and
    every `every` plies one exchange period begins:
        tick t    : sizes of the accumulated payload  -> all-gather (8 B per rank), asynchronous
        tick t+1  : sizes read (complete long ago); if any rank has records, the padded payload -> all-gather, asynchronous
        tick t+2  : payload read, records of ALL ranks returned from push()
    Nothing in a ply's loop blocks on a collective: RCCL work, its H2D / D2H staging and their completion event live on a side
    stream, and a period in which no rank finished a game costs one 8-byte all-gather and no payload step.  `flush()`
    drains the pipeline at the end (three ticks, the same on every rank).  A rank that dies mid-period does not hang the others:
    a tick that finds its collective still unfinished after `timeout_s` raises ExchangeError (the caller exits non-zero; it
    must not re-exec a process that has touched the GPU -- start a fresh child or exit)."""

    def __init__(self, device: Optional[torch.device] = None, group=None, every: int = 16, timeout_s: Optional[float] = None):
        import torch.distributed as dist

        self.group, self.every = group, max(1, int(every))
        self.timeout_s = timeout_s  # None = BETAONE_EXCHANGE_TIMEOUT (default 120 s); a collective unfinished for that long raises ExchangeError
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist_on else 1
        nccl = self.dist_on and dist.get_backend(group) == "nccl"
        self.device = torch.device(device) if (nccl and device is not None) else None
        self.side = torch.cuda.Stream(self.device) if self.device is not None else None
        self._buf: List[bytes] = []
        self._bufs: dict = {}   # staging buffers of the size gathers (int64) and of the payload gathers (uint8)
        self._calls = 0
        self._sizes = None      # (payload bytes, _Gather of the sizes)
        self._payload = None    # (sizes list, _Gather of the payload)
        self.n_size_gathers = self.n_payload_gathers = 0
        self.blocked_ticks = 0  # ticks that found their collective unfinished (should stay 0)

    def _tick(self, start_new: bool, waiting: bool = False) -> List[dict]:
        """waiting: the caller (flush) started the collective a moment ago and means to wait for it -- not a blocked tick."""
        out: List[dict] = []
        if not self.dist_on:
            if self._buf:
                out = unpack_games(b"".join(self._buf))
                self._buf = []
            return out
        if self._payload is not None:
            sizes, g = self._payload
            self.blocked_ticks += 0 if (waiting or g.done()) else 1
            rows = g.result(self.timeout_s)
            for r in range(self.world):
                out.extend(unpack_games(rows[r, :sizes[r]].tobytes()))
            self._payload = None
        if self._sizes is not None:
            payload, g = self._sizes
            self.blocked_ticks += 0 if (waiting or g.done()) else 1
            sizes = [int(x) for x in g.result(self.timeout_s).reshape(-1)]
            self._sizes = None
            mx = max(sizes)
            if mx > 0:  # every rank sees the same sizes, so every rank takes (or skips) the payload step together
                pad = np.zeros((mx + 15) // 16 * 16, dtype=np.uint8)
                pad[:len(payload)] = np.frombuffer(payload, dtype=np.uint8)
                self._payload = (sizes, _Gather(pad, self.world, self.group, self.device, self.side, self._bufs))
                self.n_payload_gathers += 1
        if start_new:
            payload = b"".join(self._buf)
            self._buf = []
            self._sizes = (payload, _Gather(np.array([len(payload)], dtype=np.int64), self.world, self.group, self.device, self.side, self._bufs))
            self.n_size_gathers += 1
        return out

    def push(self, finished: Sequence) -> List[dict]:
        self._buf.extend(pack_game(f) for f in finished)
        self._calls += 1
        return self._tick(True) if self._calls % self.every == 0 else []

    def flush(self) -> List[dict]:
        return self._tick(True) + self._tick(False, waiting=True) + self._tick(False, waiting=True)


def shard_game_ids(n_games_total: int, rank: int, world: int) -> List[int]:
    """Game id g runs on GPU g mod world; its RNG seed travels with the id."""
    return list(range(rank, n_games_total, world))
