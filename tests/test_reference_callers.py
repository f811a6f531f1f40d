"""The reference's own caller uci.py, UNMODIFIED, on top of the drop-in modules.

Runs /root/reference/uci.py as a subprocess with sys.path = [betaone_amd/dropin, oracle/shim (stands in for
python-chess, which is not installed here), ...]; the engine behind mcts.run_mcts is the wave-emulator build
(no GPU in this container).  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UCI = "/root/reference/uci.py"

BOOT = r"""
import os, sys
root = {root!r}
sys.path[:0] = [os.path.join(root, "betaone_amd", "dropin"), os.path.join(root, "oracle", "shim"), root, os.path.join(root, "tests")]
import torch
import config
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 1, 1, 32
config.NUM_SIMULATIONS = 120
import network, mcts
from engine_harness import emulator_backend
_emu = emulator_backend(); _emu.__enter__()      # no GPU here: the emulator build of the same device code, for this whole process
os.makedirs("checkpoints", exist_ok=True)
torch.manual_seed(0)
torch.save(network.PolicyValueNet().state_dict(), os.path.join("checkpoints", "stable_model(half).pth"))
import runpy
runpy.run_path({uci!r}, run_name="__main__")    # the reference file itself, unchanged
"""


@pytest.mark.skipif(not os.path.exists(UCI), reason="reference tree not present")
def test_reference_uci_runs_unchanged_on_the_dropin_modules(tmp_path):
    p = subprocess.Popen([sys.executable, "-u", "-c", BOOT.format(root=ROOT, uci=UCI)], cwd=tmp_path, stdin=subprocess.PIPE,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
    lines = []

    def send(cmd):
        p.stdin.write(cmd + "\n")
        p.stdin.flush()

    def wait_for(prefix, timeout=120):
        t0 = time.time()
        while time.time() - t0 < timeout:
            line = p.stdout.readline()
            if not line:
                if p.poll() is not None:
                    break
                continue
            lines.append(line.strip())
            if line.startswith(prefix):
                return line.strip()
        raise AssertionError(f"no {prefix!r} from uci.py; stdout={lines[-10:]} stderr={p.stderr.read()[-2000:] if p.poll() is not None else ''}")

    try:
        send("uci")
        wait_for("uciok")
        send("isready")
        wait_for("readyok")
        send("position startpos moves e2e4 e7e5 g1f3")
        wait_for("info string Position set")
        send("go movetime 300")
        best = wait_for("bestmove")
        mv = best.split()[1]
        assert len(mv) in (4, 5) and mv[0] in "abcdefgh" and mv[2] in "abcdefgh"
        send("position fen r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1")
        wait_for("info string Position set")
        send("go movetime 200")
        best2 = wait_for("bestmove")
        assert best2.split()[1] != "0000"
        send("quit")
        p.wait(timeout=30)
    finally:
        if p.poll() is None:
            p.kill()
    assert any(l.startswith("info string Search finished") for l in lines)
