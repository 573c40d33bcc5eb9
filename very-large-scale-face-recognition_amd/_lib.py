"""ctypes binding of libvlsfr.so (C-ABI declared in include/vlsfr.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or an
entry point fails, this module raises.  Build it with ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C very-large-scale-face-recognition_amd/csrc``.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p, c_size_t

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VLSFR_LIB: another build of the same library, for A/B runs of compile-time choices; the default is the in-tree build)
LIB_PATH = os.environ.get("VLSFR_LIB") or os.path.join(_HERE, "libvlsfr.so")


class VlsfrError(RuntimeError):
    pass


class DcpPlan(Structure):
    _fields_ = [("n", c_int32), ("n_ones", c_int32), ("n_special", c_int32), ("n_pos", c_int32),
                ("n_undo", c_int32), ("steps", c_int32)]


_lib = None


def _declare(lib):
    P = POINTER
    vp = c_void_p
    sig = {
        "vlsfr_last_error": (c_char_p, []),
        "vlsfr_version": (c_int, []),
        # 1. LRU
        "vlsfr_lru_create": (c_int, [c_int64, P(vp)]),
        "vlsfr_lru_destroy": (None, [vp]),
        "vlsfr_lru_get": (c_int, [vp, c_int64, P(c_int32)]),
        "vlsfr_lru_try_get": (c_int, [vp, c_int64, P(c_int32)]),
        "vlsfr_lru_view": (c_int, [vp, c_int64, P(c_int32)]),
        "vlsfr_lru_contains": (c_int, [vp, c_int64]),
        "vlsfr_lru_rollback": (c_int, [vp, c_int64, P(c_int64)]),
        "vlsfr_lru_state": (c_int, [vp, vp, vp, c_int64, P(c_int64)]),
        "vlsfr_lru_restore": (c_int, [vp, vp, vp, c_int64]),
        "vlsfr_lru_clear": (c_int, [vp]),
        "vlsfr_lru_capacity": (c_int64, [vp]),
        "vlsfr_lru_cur_idx": (c_int64, [vp]),
        "vlsfr_lru_size": (c_int64, [vp]),
        "vlsfr_lru_op_depth": (c_int64, [vp]),
        "vlsfr_lru_op_type": (c_int, [vp, c_int64]),
        # 2. DCP bookkeeping
        "vlsfr_dcp_assign": (c_int, [vp, vp, vp, vp, c_int32, c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     P(DcpPlan)]),
        "vlsfr_dcp_undo": (c_int, [vp, vp, vp, vp, P(DcpPlan)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # device entry points are declared by the modules that use them (see _declare_device)
    _declare_device(lib)


def _declare_device(lib):
    """Device entry points take raw pointers; callers pass explicit ctypes values (see head.py,
    backbone.py), so only the result types are fixed here."""
    for name in ("vlsfr_pool_scatter", "vlsfr_head_fwd_bwd"):
        getattr(lib, name).restype = c_int
    lib.vlsfr_head_workspace_bytes.restype = c_size_t


def lib():
    """The loaded library; raises VlsfrError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VlsfrError(
                "libvlsfr.so not found at %s — build it first (python -c 'import __graft_entry__ as g; "
                "g.build()'); this package has no CPU / eager fallback" % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime: load it first so that libvlsfr.so binds to the
        # same libamdhip64 (one runtime per process, shared streams and device pointers)
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
        # A/B switches from the environment, e.g. VLSFR_OPTIONS="head_variant=2,conv_glds=3" (vlsfr_set_option)
        for kv in filter(None, os.environ.get("VLSFR_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            check(_lib.vlsfr_set_option(k.strip().encode(), c_int32(int(v))), "vlsfr_set_option(%s)" % kv)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().vlsfr_last_error()
        raise VlsfrError("%s failed (%d): %s" % (what or "libvlsfr call", rc, msg.decode() if msg else "?"))
