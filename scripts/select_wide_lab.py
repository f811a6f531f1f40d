"""
scripts/select_wide_lab.py -- LAB (not part of the package): the wide synthetic workload for the PUCT-select kernel's HBM roofline
(SURVEY.md section 8d): T trees x `nodes` expanded nodes x 32 children, child blocks of 512 B
(32 records {int32 n, f32 q, f32 prior, int32 child_block}), fixed seed.  Visit counts are
Zipf-like, Q ~ U(-1,1), priors = normalised Exp(1).  All trees share one random topology (random
recursive tree) but have independent statistics, so every tree takes its own path.
"""
from __future__ import annotations

import numpy as np
import torch

from betaone_amd import engine as E

C = 32
BLOCK_I32 = 4 * C        # 128 int32 = 512 B
LEVEL_BYTES = 12 * C + 8  # N,Q,P of 32 children + child_base + parent's N


def topology(nodes: int, seed: int = 0) -> np.ndarray:
    """child_block[nodes, 32] of ONE tree (local indices, -1 = leaf): expanded nodes fill the tree level by
    level (node i's children are 32*i+1 .. 32*i+32), so with 800 nodes every descent scans 2 or 3 levels."""
    cb = np.full((nodes, C), -1, dtype=np.int32)
    for i in range(nodes):
        for c in range(C):
            j = C * i + 1 + c
            if j < nodes:
                cb[i, c] = j
    return cb


def build(n_trees: int, nodes: int = 800, seed: int = 0, device="cuda:0", n_max: int = 4096):
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    topo = torch.from_numpy(topology(nodes, seed)).to(dev)                       # [nodes, 32]
    blocks = torch.empty((n_trees * nodes, BLOCK_I32), dtype=torch.int32, device=dev)
    v = blocks.view(n_trees, nodes, BLOCK_I32)
    chunk = max(1, (1 << 28) // (nodes * C))
    for a in range(0, n_trees, chunk):
        b = min(n_trees, a + chunk)
        u = torch.rand((b - a, nodes, C), device=dev, generator=gen)
        n = (u.clamp_min(1e-6) ** (-1.0 / 1.2)).to(torch.int32).clamp_(0, n_max) - 1   # Zipf-ish, some zeros
        v[a:b, :, 0::4] = n.clamp_min_(0)
        q = torch.rand((b - a, nodes, C), device=dev, generator=gen) * 2 - 1
        v[a:b, :, 1::4] = q.view(torch.int32)
        p = -torch.log(torch.rand((b - a, nodes, C), device=dev, generator=gen).clamp_min(1e-9))
        p = p / p.sum(dim=2, keepdim=True)
        v[a:b, :, 2::4] = p.view(torch.int32)
        base = (torch.arange(a, b, device=dev, dtype=torch.int32) * nodes).view(-1, 1, 1)
        v[a:b, :, 3::4] = torch.where(topo.unsqueeze(0) >= 0, topo.unsqueeze(0) + base, torch.full_like(topo, -1).unsqueeze(0))
    root_block = (torch.arange(n_trees, device=dev, dtype=torch.int32) * nodes).contiguous()
    root_n = v[:, 0, 0::4].sum(dim=1).to(torch.int32).clamp_(0, n_max * C).contiguous()
    lut = torch.sqrt(torch.arange(n_max * C + 2, dtype=torch.float64) + 1e-8).to(torch.float32).to(dev)
    return dict(blocks=blocks, root_block=root_block, root_n=root_n, sqrt_lut=lut, n_trees=n_trees, nodes=nodes)


def run(w: dict, max_depth: int = 64, cpuct: float = 1.0, grid_blocks: int = 0, out=None):
    lib = E.load_hip_library()
    dev = w["blocks"].device
    if out is None:
        out = (torch.empty(w["n_trees"], dtype=torch.int32, device=dev), torch.empty(w["n_trees"], dtype=torch.int32, device=dev))
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.bo_select_wide(w["blocks"].data_ptr(), w["root_block"].data_ptr(), w["root_n"].data_ptr(),
                            w["sqrt_lut"].data_ptr(), w["n_trees"], max_depth, cpuct, grid_blocks,
                            out[0].data_ptr(), out[1].data_ptr(), stream)
    if rc != 0:
        raise E.EngineError(lib.bo_last_error().decode())
    return out


def reference_descent(blocks: np.ndarray, root_block: int, root_n: int, lut: np.ndarray, max_depth: int = 64,
                      cpuct: float = 1.0):
    """mcts.py:72-118 in NumPy float32 for one tree (test oracle for the wide kernel)."""
    blk, pv, pv_next, levels, leaf = int(root_block), int(root_n), int(root_n), 0, -1
    c = np.float32(cpuct)
    while blk >= 0 and levels < max_depth:
        row = blocks[blk]
        n = row[0::4]
        q = row[1::4].view(np.float32)
        p = row[2::4].view(np.float32)
        sp = lut[pv]
        best, bi = -np.inf, 0
        for i in range(C):
            t2 = np.float32(np.float32(c * p[i]) * sp)
            score = np.float32(q[i] + np.float32(t2 / np.float32(1 + n[i]))) if n[i] > 0 else np.float32(np.float32(0.0) + t2)
            if score > best:
                best, bi = score, i
        leaf = blk * C + bi
        pv, pv_next = pv_next, int(n[bi])
        blk = int(row[3 + 4 * bi])
        levels += 1
    return leaf, levels
