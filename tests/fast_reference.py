"""tests/fast_reference.py -- NumPy/oracle restatement of the FAST search mode (betaone_amd/csrc/bo_fastw.h).

There is no reference implementation of this mode (it deliberately diverges from BetaOne's search), so the kernels are
checked against this independent, readable restatement: same selection rule, same virtual loss, same step structure
(L descents, then the leaves' positions, then expansion and backup), same summation order (lane-strided partial sums +
butterfly), same float32 operation order, same tree reuse between moves.  Rules/encoding come from the oracle.  It keeps
an ordinary pointer tree -- nothing of the engine's child-block arenas -- and is compared through a path-keyed canonical
form."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O

F = np.float32
PATH_CAP = 64
UNVISITED, MATE, DRAW, EXPANDED = "unvisited", "mate", "draw", "expanded"


def wave_sum(vals):
    part = np.zeros(64, dtype=np.float32)
    for j, v in enumerate(vals):
        part[j % 64] = F(part[j % 64] + F(v))
    lanes = np.arange(64)
    for m in (1, 2, 4, 8, 16, 32):
        part = (part + part[lanes ^ m]).astype(np.float32)
    return F(part[0])


class Node:
    __slots__ = ("n", "w", "prior", "move", "state", "children", "row", "fl")

    def __init__(self, prior, move):
        self.n, self.w, self.prior, self.move = 0, F(0.0), F(prior), move
        self.state, self.children, self.row = UNVISITED, [], -1
        self.fl = 0  # descents of the step in flight through the node


class FastSearcher:
    """One game: search() runs `sims` simulations from the current position, play(uci) advances the game and keeps the
    played child's subtree (tree reuse) unless reuse=False."""

    def __init__(self, board: O.Board, trk, eval_fn, sims, L, cpuct=1.0, eps=0.25, use_noise=True, reuse=True):
        self.board, self.trk, self.eval_fn, self.sims, self.L = board, trk, eval_fn, sims, L
        self.cp, self.keep, self.eps, self.use_noise, self.reuse = F(cpuct), F(1.0 - eps), eps, use_noise, reuse
        self.root = Node(1.0, None)
        self.n_evals = self.n_term_sims = 0

    # -- helpers ------------------------------------------------------------------------------------------------------
    def _hist(self):
        pos = self.board.positions()
        return pos[max(0, len(pos) - 8):-1]

    def _materialise(self, path_moves):
        """-> (termination code 0/1/2, legal moves in python-chess order, planes)"""
        b = self.board
        hist = self._hist()
        for m in path_moves:
            b.push(O.move_from_uci(m))
        t = b.termination()
        t = 0 if t == 0 else (1 if t == 1 else 2)
        moves = [O.move_to_uci(m) for m in b.legal_moves()]
        planes = O.encode_board(list(hist) + [b.pos.copy()], self.trk) if t == 0 else None
        for _ in path_moves:
            b.pop()
        return t, moves, planes

    def _expand(self, node, moves, probs_row, noise):
        pv = [F(probs_row[O.move_to_index(O.move_from_uci(m))]) for m in moves]
        s = wave_sum(pv)
        pv = [F(p / s) if s > 0 else F(F(1.0) / F(len(pv))) for p in pv]
        if noise is not None:
            pv = [F(np.float64(F(self.keep * p)) + self.eps * noise[j]) for j, p in enumerate(pv)]
        node.children = [Node(p, m) for m, p in zip(moves, pv)]
        node.state = EXPANDED

    @staticmethod
    def _backup(path, v):
        plen = len(path)
        path[0].n += 1
        path[0].fl = 0
        for k in range(1, plen):
            nd = path[k]
            s = F(-v) if ((plen - 1 - k) & 1) else F(v)
            nd.n += 1
            nd.w = F(nd.w + s)
            nd.fl = 0

    # -- one search -------------------------------------------------------------------------------------------------------
    def search(self, noise):
        root, done = self.root, 0
        t_root, root_moves, root_planes = self._materialise([])
        if t_root != 0:
            return
        noise = noise if self.use_noise else None
        if root.state != EXPANDED:  # its evaluation is a step of its own (one row, no simulation)
            p, _v = self.eval_fn(root_planes[None])
            self.n_evals += 1
            self._expand(root, root_moves, p[0], noise)
            root.n = 1
        elif noise is not None:  # a root kept from the previous search gets its Dirichlet noise now
            for j, ch in enumerate(root.children):
                ch.prior = F(np.float64(F(self.keep * ch.prior)) + self.eps * noise[j])
        while done < self.sims:
            rows, sims_l = [], []  # rows: leaf nodes + their path moves; sims_l: (path, row | "mate" | "draw")
            while len(sims_l) < self.L and done + len(sims_l) < self.sims:
                root.fl += 1
                path, cur = [root], root
                while cur.state == EXPANDED and len(path) < PATH_CAP:
                    sq = np.sqrt(F(cur.n + cur.fl))  # visits of the node being expanded, this simulation included
                    best, bi = -np.inf, None
                    for ch in cur.children:
                        ne = ch.n + ch.fl
                        we = F(ch.w - F(ch.fl))
                        t2 = F(F(self.cp * ch.prior) * sq)
                        u = F(t2 * F(F(1.0) / F(1 + ne)))
                        qv = F(we * F(F(1.0) / F(ne))) if ne > 0 else F(0.0)
                        sc = F(qv + u)
                        if sc > best:
                            best, bi = sc, ch
                    cur = bi
                    cur.fl += 1
                    path.append(cur)
                assert cur.state != EXPANDED, "path cap not expected in tests"
                if cur.state in (MATE, DRAW):  # known terminal: needs no row; backed up with the step's other simulations
                    sims_l.append((path, cur.state))
                elif cur.row >= 0:  # already selected in this step: shares the row
                    sims_l.append((path, cur.row))
                else:
                    cur.row = len(rows)
                    rows.append((cur, [nd.move for nd in path[1:]]))
                    sims_l.append((path, cur.row))
            info, vals = [], {}
            if rows:
                info = [self._materialise(pm) for _leaf, pm in rows]
                live = [i for i, (t, _m, _p) in enumerate(info) if t == 0]
                if live:
                    probs, v = self.eval_fn(np.stack([info[i][2] for i in live]))
                    self.n_evals += len(live)
                    for k, i in enumerate(live):
                        vals[i] = (probs[k], F(v[k]))
                for i, (leaf, _pm) in enumerate(rows):
                    t, moves, _planes = info[i]
                    leaf.row = -1
                    if t:
                        leaf.state = MATE if t == 1 else DRAW
                    else:
                        self._expand(leaf, moves, vals[i][0], None)
            for path, r in sims_l:
                if r in (MATE, DRAW):
                    self.n_term_sims += 1
                    self._backup(path, F(1.0) if r == MATE else F(0.0))
                else:
                    t = info[r][0]
                    if t:
                        self.n_term_sims += 1
                    self._backup(path, (F(1.0) if t == 1 else F(0.0)) if t else F(-vals[r][1]))
            done += len(sims_l)

    def visits(self):
        return [(ch.move, ch.n) for ch in self.root.children]

    def play(self, uci):
        child = next((ch for ch in self.root.children if ch.move == uci), None)
        self.board.push(O.move_from_uci(uci))
        self.trk.add_board(self.board)
        if self.reuse and child is not None and child.state == EXPANDED:
            child.w, child.prior, child.move = F(0.0), F(1.0), None
            self.root = child
        else:
            self.root = Node(1.0, None)

    def canonical(self):
        """{move path: (n, w bits, prior bits, n_children, terminal code)}; the root's own w / prior are not compared."""
        out = {}

        def walk(nd, path):
            term = 1 if nd.state == MATE else 2 if nd.state == DRAW else 0 if nd.state == EXPANDED else -1
            wb = None if not path else int(np.float32(nd.w).view(np.uint32))
            pb = None if not path else int(np.float32(nd.prior).view(np.uint32))
            out["/".join(path)] = (int(nd.n), wb, pb, len(nd.children), term)
            for ch in nd.children:
                walk(ch, path + [ch.move])

        walk(self.root, [])
        return out


def canonical_from_engine(nodes, move_to_uci):
    """The same canonical form from Engine.debug_tree() of a fast-mode engine."""
    paths, out = {}, {}
    for i, nd in enumerate(nodes):
        path = [] if nd["parent"] < 0 else paths[nd["parent"]] + [move_to_uci(nd["move"])]
        paths[i] = path
        wb = None if not path else int(np.float32(nd["q"]).view(np.uint32))
        pb = None if not path else int(np.float32(nd["prior"]).view(np.uint32))
        out["/".join(path)] = (int(nd["n"]), wb, pb, int(nd["n_children"]), int(nd["terminal"]))
    return out
