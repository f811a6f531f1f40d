"""betaone_amd/build.py -- compiles csrc/ into csrc/libbetaone_hip.so with hipcc for gfx950 (in-tree)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libbetaone_hip.so")
SOURCES = ["bo_engine.cpp"]
HEADERS = ["bo_wave.h", "bo_chess.h", "bo_tree.h", "bo_select_wide.h", "bo_rt.h", "../../include/betaone_engine.h"]
# strict IEEE binary32 in the tree arithmetic (parity with the reference's NumPy float32 scalars):
# no FMA contraction, correctly rounded fp32 divide, denormals kept.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-x", "hip"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
