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


def test_head_cfg_mirror_matches_library_struct():
    """The ctypes mirror of vlsfr_head_cfg (head.py, and the stub in INTEGRATION.md section 3) has the size the
    library was compiled with, and the documented stub lists exactly the header's fields in order."""
    from vlsfr_amd.head import HeadCfg
    L = _lib.lib()
    L.vlsfr_head_cfg_size.restype = ctypes.c_size_t
    assert L.vlsfr_head_cfg_size() == ctypes.sizeof(HeadCfg)
    hdr = open(os.path.join(ROOT, "include", "vlsfr.h")).read()
    body = hdr[hdr.index("typedef struct vlsfr_head_cfg {"):hdr.index("} vlsfr_head_cfg;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([A-Za-z_0-9]+)\s*;", body)
    assert fields == [f[0] for f in HeadCfg._fields_]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class HeadCfg(ctypes.Structure):"):]
    stub = stub[:stub.index("]\n") + 1]
    assert re.findall(r'\("([A-Za-z_0-9]+)",', stub) == fields
