"""CPU: the product library loads and exports exactly the C ABI include/gamma_hip.h declares.
No compute calls (no GPU here); gamma_hip_create must fail cleanly without a device."""
import ctypes as C
import os
import re

import pytest

from gamma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gamma_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gamma_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libgamma_hip.so not built (run __graft_entry__.build())")
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    assert sorted(_lib.SYMBOLS) == names, "python binding table out of sync with the header"


def test_create_fails_loudly_without_gpu():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libgamma_hip.so not built")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gamma_amd import api
    with pytest.raises(api.GammaHipError):
        api.GammaHip(0)


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgamma_hip.so")
    with pytest.raises(_lib.GammaHipError):
        _lib.load()


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gamma_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".cc")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", src), f
                assert "libgamma_oracle" not in src, f
                # the one library the product resolves at run time is RCCL (the group's optional transport)
                for m in re.finditer(r"dlopen\s*\(\s*([^,)]*)", src):
                    assert m.group(1).strip().startswith('"librccl.so'), (f, m.group(0))
