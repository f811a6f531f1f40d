"""tests/fast_reference.py -- NumPy/oracle restatement of the FAST search mode (betaone_amd/csrc/bo_fast.h).

There is no reference implementation of this mode (it deliberately diverges from BetaOne's search), so the kernels
are checked against this independent, readable restatement: same selection rule, same virtual loss, same summation
order (lane-strided partial sums + butterfly), same float32 operation order.  Rules/encoding come from the oracle."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O

F = np.float32
PATH_CAP = 192


def wave_sum(vals):
    part = np.zeros(64, dtype=np.float32)
    for j, v in enumerate(vals):
        part[j % 64] = F(part[j % 64] + F(v))
    lanes = np.arange(64)
    for m in (1, 2, 4, 8, 16, 32):
        part = (part + part[lanes ^ m]).astype(np.float32)
    return F(part[0])


class Node:
    __slots__ = ("n", "w", "prior", "parent", "first", "nc", "move", "term", "moves", "pending")

    def __init__(self, parent, prior, move):
        self.n, self.w, self.prior, self.parent, self.first, self.nc = 0, F(0.0), F(prior), parent, 0, 0
        self.move, self.term, self.moves, self.pending = move, -1, None, -1


def fast_search(board: O.Board, hist, trk, eval_fn, noise, sims, L, cpuct=1.0, eps=0.25, use_noise=True):
    """Returns the node list (creation order).  eval_fn(planes[n,120,8,8]) -> (probs[n,4672], values[n])."""
    nodes = [Node(-1, 1.0, None)]
    cp, keep = F(cpuct), F(1.0 - eps)

    def path_moves(i):
        out = []
        while i > 0:
            out.append(nodes[i].move)
            i = nodes[i].parent
        return out[::-1]

    def with_board(i):
        mv = path_moves(i)
        for m in mv:
            board.push(O.move_from_uci(m))
        return len(mv)

    def unwind(k):
        for _ in range(k):
            board.pop()

    def materialise(i):
        k = with_board(i)
        t = board.termination()
        nodes[i].term = 0 if t == 0 else (1 if t == 1 else 2)
        nodes[i].moves = [O.move_to_uci(m) for m in board.legal_moves()]
        planes = O.encode_board(list(hist) + [board.pos.copy()], trk)
        unwind(k)
        return planes

    def backup(path, v):
        plen = len(path)
        for k in range(1, plen):
            nd = nodes[path[k]]
            s = F(-v) if ((plen - 1 - k) & 1) else F(v)
            nd.w = F(F(nd.w + F(1.0)) + s)

    def expand(i, probs_row, is_root):
        nd = nodes[i]
        pv = [F(probs_row[O.move_to_index(O.move_from_uci(m))]) for m in nd.moves]
        s = wave_sum(pv)
        pv = [F(p / s) if s > 0 else F(F(1.0) / F(len(pv))) for p in pv]
        if is_root and use_noise:
            pv = [F(np.float64(F(keep * p)) + eps * noise[j]) for j, p in enumerate(pv)]
        nd.first, nd.nc = len(nodes), len(pv)
        for m, p in zip(nd.moves, pv):
            nodes.append(Node(i, p, m))
        nd.pending = -1

    root_planes = materialise(0)
    done = 0
    if nodes[0].term == 0:
        p, _v = eval_fn(root_planes[None])
        expand(0, p[0], True)
        nodes[0].n = 1
        while done < sims:
            rows, sims_l = [], []          # rows: (leaf, planes); sims_l: (path, row or -1)
            while len(sims_l) < L and done + len(sims_l) < sims:
                path, cur = [0], 0
                nodes[0].n += 1
                while nodes[cur].nc > 0 and len(path) < PATH_CAP:
                    par = nodes[cur]
                    sq = np.sqrt(F(par.n))
                    best, bi = -np.inf, 0
                    for i in range(par.nc):
                        ch = nodes[par.first + i]
                        t2 = F(F(cp * ch.prior) * sq)
                        u = F(t2 / F(1 + ch.n))
                        qv = F(ch.w / F(ch.n)) if ch.n > 0 else F(0.0)
                        sc = F(qv + u)
                        if sc > best:
                            best, bi = sc, i
                    cur = par.first + bi
                    path.append(cur)
                    nodes[cur].n += 1
                    nodes[cur].w = F(nodes[cur].w - F(1.0))
                leaf = nodes[cur]
                row = leaf.pending
                if leaf.moves is None:
                    planes = materialise(cur)
                    if leaf.term == 0:
                        leaf.pending = row = len(rows)
                        rows.append((cur, planes))
                elif leaf.term == 0 and row < 0 and leaf.nc == 0:
                    raise AssertionError("re-evaluation path not expected in tests")
                if leaf.term > 0:
                    backup(path, F(1.0) if leaf.term == 1 else F(0.0))
                    sims_l.append((path, -1))
                else:
                    sims_l.append((path, row))
            if rows:
                probs, vals = eval_fn(np.stack([pl for _, pl in rows]))
                for r, (leaf_i, _) in enumerate(rows):
                    expand(leaf_i, probs[r], False)
                for path, r in sims_l:
                    if r >= 0:
                        backup(path, F(-vals[r]))
            done += len(sims_l)
    return nodes
