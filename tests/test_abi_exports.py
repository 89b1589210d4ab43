"""CPU: the gfx950 library builds (hipcc cross-compiles without a GPU), loads, and exports every
entry point `include/i2v_hip.h` declares; the ctypes table in `i2v_amd/lib.py` covers the same set.
No compute calls are made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "i2v_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(i2v_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from i2v_amd import lib
    cd = ctypes.CDLL(lib.LIB_PATH)
    names = declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(cd, n)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == names
    assert cd.i2v_abi_version() == 1
    cd.i2v_backend.restype = ctypes.c_char_p
    assert cd.i2v_backend() == b"hip:gfx950"


def test_product_refuses_anything_but_the_hip_build(monkeypatch):
    """No CPU fallback: a missing library or a non-HIP backend is an error, not a detour."""
    from i2v_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", os.path.join(ROOT, "tests", "hostsim", "libi2v_hostsim.so"))
    if os.path.exists(lib.LIB_PATH):
        with pytest.raises(lib.I2VError, match="backend"):
            lib.load()
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libi2v_hip.so")
    with pytest.raises(lib.I2VError, match="no CPU fallback"):
        lib.load()
    from i2v_amd.engine import Engine
    with pytest.raises(lib.I2VError):
        Engine("cpu")
