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
    ptr = np.zeros(n + 1, dtype=np.int32)
    for i, (idx, _) in enumerate(fin.pis):
        ptr[i + 1] = ptr[i] + len(idx)
    idx = np.concatenate([np.asarray(i, dtype=np.int32) for i, _ in fin.pis]) if n else np.zeros(0, np.int32)
    val = np.concatenate([np.asarray(v, dtype=np.float32) for _, v in fin.pis]) if n else np.zeros(0, np.float32)
    head = np.array([MAGIC, fin.game_id, n, fin.terminal, 0, len(idx)], dtype=np.int32)
    head[4:5].view(np.float32)[0] = fin.outcome
    pos = b"".join(bytes(p) for p in fin.positions[:n + 1])
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


class LaggedGameExchange:
    """all_gather_games with the size exchange one step behind: `push(finished)` starts the (tiny) size all-gather of this
    step's records asynchronously and completes the PREVIOUS step's exchange, whose sizes arrived while a whole ply ran on
    the GPU -- the host never waits for a collective it has just issued.  Every rank must call push() once per step and
    flush() once at the end; records are delivered one step late."""

    def __init__(self, device: Optional[torch.device] = None, group=None):
        self.device, self.group, self._pending = device, group, None

    def _start(self, finished):
        import torch.distributed as dist

        payload = b"".join(pack_game(f) for f in finished)
        if not dist.is_available() or not dist.is_initialized():
            return (payload, None, None)
        world = dist.get_world_size(self.group)
        backend = dist.get_backend(self.group)
        dev = self.device if (self.device is not None and backend == "nccl") else torch.device("cpu")
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        mine = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
        work = dist.all_gather_into_tensor(sizes, mine, group=self.group, async_op=True)
        return (payload, sizes, (work, mine, dev))

    def _finish(self, pending) -> List[dict]:
        import torch.distributed as dist

        payload, sizes, h = pending
        if h is None:
            return unpack_games(payload) if payload else []
        work, _mine, dev = h
        work.wait()
        sizes_h = sizes.cpu().tolist()
        mx = max(sizes_h)
        if mx == 0:
            return []
        pad = torch.zeros(mx, dtype=torch.uint8)
        if payload:
            pad[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        pad = pad.to(dev)
        gathered = torch.empty(len(sizes_h) * mx, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, pad, group=self.group)
        g = gathered.cpu().numpy()
        games: List[dict] = []
        for r in range(len(sizes_h)):
            games.extend(unpack_games(g[r * mx:r * mx + sizes_h[r]].tobytes()))
        return games

    def push(self, finished: Sequence) -> List[dict]:
        nxt = self._start(finished)
        out = self._finish(self._pending) if self._pending is not None else []
        self._pending = nxt
        return out

    def flush(self) -> List[dict]:
        out = self._finish(self._pending) if self._pending is not None else []
        self._pending = None
        return out


def shard_game_ids(n_games_total: int, rank: int, world: int) -> List[int]:
    """Game id g runs on GPU g mod world; its RNG seed travels with the id."""
    return list(range(rank, n_games_total, world))
