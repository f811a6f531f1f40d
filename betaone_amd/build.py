"""betaone_amd/build.py -- compiles csrc/ into csrc/libbetaone_hip.so with hipcc for gfx950 (in-tree)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libbetaone_hip.so")
SOURCES = ["bo_engine.cpp"]
# strict IEEE binary32 in the tree arithmetic (parity with the reference's NumPy float32 scalars):
# no FMA contraction, correctly rounded fp32 divide, denormals kept.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-x", "hip"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def dependencies():
    """Everything libbetaone_hip.so is compiled from: every file of csrc/ (bo_engine.cpp includes all the headers), the
    public headers, and this file (the flags)."""
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".cpp", ".hip"))]
    inc = os.path.join(HERE, "..", "include")
    return deps + [os.path.join(inc, "betaone_engine.h"), os.path.join(inc, "betaone_lab.h"), os.path.abspath(__file__)]


def source_hash() -> str:
    import hashlib

    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in dependencies()[:-1]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build() -> bool:
    """Stale = the library is missing or was not built from the sources next to it (content hash kept in a sidecar file:
    the .so is git-ignored but travels to the GPU box, where file times say nothing)."""
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".srchash"):
        return True
    return open(LIB + ".srchash").read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(LIB + ".srchash", "w") as f:
            f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
