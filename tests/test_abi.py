"""CPU test: the product library loads and exports every symbol include/betaone_engine.h declares
(no compute calls -- there is no GPU here)."""
import ctypes as C
import os
import re

from betaone_amd import build, engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "betaone_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bo_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(E._SYMBOLS)


def test_hip_library_builds_loads_and_exports_the_abi():
    lib_path = build.build()
    lib = C.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    E.bind(lib)
    assert lib.bo_abi_version() == 1


def test_product_loader_has_no_fallback(monkeypatch, tmp_path):
    import pytest

    monkeypatch.setattr(E, "HIP_LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(E, "_hip_lib", None)
    with pytest.raises(E.EngineError):
        E.load_hip_library()
