"""The C-ABI library loads and exports every symbol include/vlsfr.h declares (no compute calls)."""
import ctypes
import os
import re

from vlsfr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vlsfr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vlsfr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) > 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_error_string():
    L = _lib.lib()
    assert L.vlsfr_version() >= 100
    out = ctypes.c_void_p()
    assert L.vlsfr_lru_create(0, ctypes.byref(out)) == -1
    assert b"capacity" in L.vlsfr_last_error()
