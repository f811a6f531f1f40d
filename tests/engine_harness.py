"""tests/engine_harness.py -- drives betaone_amd.engine.Engine for the parity tests.

backend "hip": the product library (csrc/libbetaone_hip.so) on cuda:0, buffers are torch CUDA tensors.
backend "emu": the SAME device sources compiled against tests/wave_emulator (CPU, 64-lane lockstep
               emulator), buffers are numpy arrays.  Test infrastructure only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from betaone_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "wave_emulator")
EMU_LIB = os.path.join(EMU_DIR, "libbetaone_emu.so")
_emu = None


def emu_lib():
    global _emu
    if _emu is None and os.environ.get("BO_EMU_LIB"):  # e.g. an AddressSanitizer build (tests/wave_emulator/README)
        _emu = E.bind(C.CDLL(os.environ["BO_EMU_LIB"]))
    if _emu is None:
        srcs = [os.path.join(ROOT, "betaone_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "betaone_amd", "csrc"))
                if f.endswith((".h", ".cpp"))] + [os.path.join(EMU_DIR, "wave_emu.h"),
                                                  os.path.join(ROOT, "include", "betaone_engine.h"), os.path.join(ROOT, "include", "betaone_lab.h")]
        if not os.path.exists(EMU_LIB) or any(os.path.getmtime(s) > os.path.getmtime(EMU_LIB) for s in srcs):
            subprocess.check_call([os.path.join(EMU_DIR, "build.sh")])
        _emu = E.bind(C.CDLL(EMU_LIB))
    return _emu


import contextlib


@contextlib.contextmanager
def emulator_backend():
    """Test infrastructure: inside this block betaone_amd.engine hands out the wave-emulator build of the device code and a
    CPU torch device, so kernel logic runs in a GPU-less container.  The product has no such switch -- this patches the
    two module functions from the outside and restores them."""
    import torch

    saved = (E.load_hip_library, E.runtime_device)
    E.load_hip_library = emu_lib
    E.runtime_device = lambda requested: torch.device("cpu")
    try:
        yield
    finally:
        E.load_hip_library, E.runtime_device = saved


def emu_call(fn, *a, **kw):
    with emulator_backend():
        return fn(*a, **kw)


class Buf:
    """float32 device buffer with a raw address."""

    def __init__(self, backend: str, shape):
        self.backend = backend
        if backend == "emu":
            self.a = np.zeros(shape, dtype=np.float32)
            self.ptr = self.a.ctypes.data
        else:
            import torch

            self.t = torch.zeros(shape, dtype=torch.float32, device="cuda:0")
            self.ptr = self.t.data_ptr()

    def numpy(self) -> np.ndarray:
        if self.backend == "emu":
            return self.a
        import torch

        torch.cuda.synchronize()
        return self.t.cpu().numpy()

    def set(self, arr: np.ndarray):
        if self.backend == "emu":
            self.a[...] = arr
        else:
            import torch

            self.t.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)))
            torch.cuda.synchronize()


def make_engine(backend: str, n_games: int, cfg: dict, max_plies: int = 512) -> E.Engine:
    kw = dict(num_simulations=cfg.get("num_simulations", 250), mcts_batch_size=cfg.get("batch_size", 96),
              cpuct=cfg.get("cpuct", 1.0), widen_coeff=cfg.get("widen_coeff", 1.5),
              dirichlet_alpha=cfg.get("dirichlet_alpha", 0.1), dirichlet_epsilon=cfg.get("dirichlet_eps", 0.25),
              max_plies=max_plies)
    if backend == "emu":
        return emu_call(E.Engine, n_games, **kw)
    return E.Engine(n_games, **kw)


class Searcher:
    """Runs searches with an EXTERNAL evaluator (planes -> softmax probs, value), i.e. from the
    (priors, value) seam of SURVEY.md section 8c onward the engine sees exactly what the reference saw."""

    def __init__(self, backend: str, eng: E.Engine):
        self.backend, self.eng, self.G = backend, eng, eng.G
        self.nn_in = Buf(backend, (self.G, 120, 8, 8))
        self.policy = Buf(backend, (self.G, E.NUM_ACTIONS))
        self.value = Buf(backend, (self.G,))
        self.n_eval_rows = 0
        self.n_steps = 0

    def search(self, go, eval_fns, rngs, alpha: float):
        """go[g]: run a search in slot g; eval_fns[g](planes[n,120,8,8]) -> (probs, values);
        rngs[g]: numpy RandomState standing in for the reference's global RNG (mcts.py:192)."""
        eng, G = self.eng, self.G
        nl, term, _ = eng.root_info()
        noise = np.zeros((G, E.MAX_LEGAL), dtype=np.float64)
        for g in range(G):
            if go[g] and term[g] == 0 and alpha > 0:
                noise[g, :nl[g]] = rngs[g].dirichlet([alpha] * int(nl[g]))
        eng.search_begin(go, noise if alpha > 0 else None, self.nn_in.ptr)
        kind = E.POLICY_NONE
        while True:
            eng.step(self.policy.ptr, self.value.ptr, kind, self.nn_in.ptr)
            self.n_steps += 1
            running, requested, mask = eng.poll()
            if running == 0:
                break
            assert requested <= running  # (a game absorbing a long run of terminal simulations yields without a request)
            planes = self.nn_in.numpy()
            pol = np.zeros((G, E.NUM_ACTIONS), dtype=np.float32)
            val = np.zeros(G, dtype=np.float32)
            for g in np.nonzero(mask)[0]:
                p, v = eval_fns[g](planes[g:g + 1])
                pol[g], val[g] = p[0], v[0]
                self.n_eval_rows += 1
            self.policy.set(pol)
            self.value.set(val)
            kind = E.POLICY_PROBS
        eng.check_status()
        return eng.result()


def canonical_tree(nodes) -> dict:
    """{move path: [n, q bits (None for the root), prior bits, n_children]} -- same form as the fixtures."""
    paths, out = {}, {}
    for i, nd in enumerate(nodes):
        path = [] if nd["parent"] < 0 else paths[nd["parent"]] + [E.move_to_uci(nd["move"])]
        paths[i] = path
        qb = None if i == 0 else int(np.float32(nd["q"]).view(np.uint32))
        out["/".join(path)] = [int(nd["n"]), qb, int(np.float32(nd["prior"]).view(np.uint32)), int(nd["n_children"])]
    return out


def dense_pi(res, g: int) -> np.ndarray:
    pi = np.zeros(E.NUM_ACTIONS, dtype=np.float32)
    n = int(res["n"][g])
    pi[res["idx"][g, :n]] = res["val"][g, :n]
    return pi
