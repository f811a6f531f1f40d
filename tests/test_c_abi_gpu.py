"""GPU test: a plain-C program (tests/c_abi_smoke.c) linked against libbetaone_hip.so drives the engine through
include/betaone_engine.h only -- no Python binding, no torch types."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_drives_the_engine(tmp_path):
    exe = str(tmp_path / "c_abi_smoke")
    lib_dir = os.path.join(ROOT, "betaone_amd", "csrc")
    # plain gcc: the consumer only needs the C header, the library and (for its own buffers) the HIP runtime API
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c_abi_smoke.c"),
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-L", lib_dir, "-lbetaone_hip",
                           "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI smoke ok" in out.stdout
