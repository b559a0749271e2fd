"""The oracle is test infrastructure: nothing under deformablelka_amd/ (the product) or in the C-ABI sources may import,
link or execute it, and the product must fail loudly — not fall back — when the HIP library is missing."""
import ast
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deformablelka_amd")


def _py_files():
    for d, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_imports_oracle():
    bad = []
    for path in _py_files():
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                if n.split(".")[0] in ("oracle", "tests"):
                    bad.append((path, n))
    assert not bad, bad


def test_native_sources_do_not_reference_oracle():
    csrc = os.path.join(PKG, "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            txt = open(os.path.join(csrc, f)).read()
            assert not re.search(r"oracle/|dlka_oracle", txt), f


def test_missing_library_fails_loudly(monkeypatch):
    from deformablelka_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does", "not", "exist.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.get_lib()


def test_cpu_tensors_are_rejected_by_the_product_backend():
    import torch
    from deformablelka_amd import _lib
    if _lib._test_backend:
        pytest.skip("emulator backend active")
    with pytest.raises(RuntimeError, match="not implemented on the CPU"):
        _lib.require_device(torch.zeros(1))


def test_every_declared_symbol_is_exported():
    """include/dlka.h <-> libdlka_hip.so <-> _lib.SIGNATURES (no compute calls: there is no GPU here)."""
    from deformablelka_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dlka.h")).read()
    declared = set(re.findall(r"\b(dlka_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dlka_status", "dlka_dtype"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdlka_hip.so not built")
    import ctypes
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), name
    assert cdll.dlka_abi_version() == 1
