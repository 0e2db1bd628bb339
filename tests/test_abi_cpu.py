"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(xva_[a-z0-9_]+)\s*\(", src):
            syms.add(m.group(1))
    return syms


def test_library_loads_and_exports_every_declared_symbol():
    from xva_trainer_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert "xva_gemm" in syms and "xva_mel_spectrogram" in syms
    missing = [s for s in sorted(syms) if not hasattr(_lib.lib, s)]
    assert not missing, "declared in include/ but not exported: %s" % missing
    assert _lib.lib.xva_abi_version() >= 1
    assert _lib.lib.xva_target_arch() == b"gfx950"


def test_no_cpu_fallback_ops_refuse_host_tensors():
    import pytest
    import torch
    from xva_trainer_amd import _lib
    with pytest.raises(_lib.XvaError):
        _lib.require_cuda(torch.zeros(4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "xva-trainer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)


def test_error_string_roundtrip():
    from xva_trainer_amd import _lib
    p = _lib.GemmParams()
    rc = _lib.lib.xva_gemm(ctypes.byref(p), None)
    assert rc < 0 and b"null" in _lib.lib.xva_last_error()
