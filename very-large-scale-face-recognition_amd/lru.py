"""LRU slot allocator of the Dynamic Class Pool — Python face of the native allocator.

Same public surface as the reference ``lru.LRU`` (lru.py:21-255): ``LRU(capacity)``, ``get``,
``try_get``, ``view``, ``key in lru``, ``rollback_one_step``, ``rollback_steps``, ``state_dict``,
``restore``, ``clear``, ``keys``, iteration, and the public fields ``capacity``, ``cur_idx``,
``cache``, ``op_stack``.  All state lives in libvlsfr.so (csrc/lru.cpp); this class only forwards.
"""
import ctypes
from collections.abc import Mapping, Sequence

import numpy as np

from . import _lib

_OP_NAMES = ("Add", "Overflow", "Get")


def _as_key(key):
    if isinstance(key, (bool, np.bool_)) or not isinstance(key, (int, np.integer)):
        raise TypeError("LRU keys are integer identity labels, got %r" % (type(key).__name__,))
    return int(key)


class _CacheView(Mapping):
    """Read-only stand-in for the reference's ``LRU.cache`` dict (key -> slot)."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return int(_lib.lib().vlsfr_lru_size(self._o._h))

    def __contains__(self, key):
        return key in self._o

    def __getitem__(self, key):
        v = self._o.view(key)
        if v < 0:
            raise KeyError(key)
        return v

    def __iter__(self):
        return (k for k, _ in self._o.state_dict())


class _OpRecord(object):
    __slots__ = ("op_type",)

    def __init__(self, op_type):
        self.op_type = op_type


class _OpStackView(Sequence):
    """Read-only stand-in for ``LRU.op_stack`` (len and ``[i].op_type``)."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return int(_lib.lib().vlsfr_lru_op_depth(self._o._h))

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return _OpRecord(_OP_NAMES[_lib.lib().vlsfr_lru_op_type(self._o._h, i)])


class LRU(object):
    def __init__(self, capacity):
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(self._L.vlsfr_lru_create(int(capacity), ctypes.byref(h)), "vlsfr_lru_create")
        self._h = h
        self.capacity = int(capacity)
        self.cache = _CacheView(self)
        self.op_stack = _OpStackView(self)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and self._L is not None:
            self._L.vlsfr_lru_destroy(h)
            self._h = None

    @property
    def cur_idx(self):
        return int(self._L.vlsfr_lru_cur_idx(self._h))

    # lru.py:44-89
    def get(self, key):
        out = ctypes.c_int32()
        _lib.check(self._L.vlsfr_lru_get(self._h, _as_key(key), ctypes.byref(out)), "vlsfr_lru_get")
        return out.value

    # lru.py:157-204
    def try_get(self, key):
        out = ctypes.c_int32()
        _lib.check(self._L.vlsfr_lru_try_get(self._h, _as_key(key), ctypes.byref(out)), "vlsfr_lru_try_get")
        return out.value

    # lru.py:147-151
    def view(self, key):
        out = ctypes.c_int32()
        _lib.check(self._L.vlsfr_lru_view(self._h, _as_key(key), ctypes.byref(out)), "vlsfr_lru_view")
        return out.value

    # lru.py:145
    def __contains__(self, key):
        if isinstance(key, (bool, np.bool_)) or not isinstance(key, (int, np.integer)):
            return False
        return bool(self._L.vlsfr_lru_contains(self._h, int(key)))

    # lru.py:210-248
    def rollback_one_step(self):
        _lib.check(self._L.vlsfr_lru_rollback(self._h, 1, None), "vlsfr_lru_rollback")

    # lru.py:252-255
    def rollback_steps(self, steps):
        _lib.check(self._L.vlsfr_lru_rollback(self._h, max(int(steps), 0), None), "vlsfr_lru_rollback")

    def _state_arrays(self):
        n = int(self._L.vlsfr_lru_size(self._h))
        keys = np.empty(n, dtype=np.int64)
        slots = np.empty(n, dtype=np.int32)
        got = ctypes.c_int64()
        _lib.check(self._L.vlsfr_lru_state(self._h, keys.ctypes.data, slots.ctypes.data, n, ctypes.byref(got)),
                   "vlsfr_lru_state")
        return keys[:got.value], slots[:got.value]

    # lru.py:102-108 — MRU -> LRU list of (key, slot)
    def state_dict(self):
        keys, slots = self._state_arrays()
        return list(zip(keys.tolist(), slots.tolist()))

    def __iter__(self):
        return iter(self.state_dict())

    def keys(self):
        return self.cache.keys()

    # lru.py:113-128 — the reference asserts; so do we
    def restore(self, kvs):
        kvs = list(kvs)
        assert len(kvs) <= self.capacity
        assert self.cur_idx == 0
        keys = np.asarray([_as_key(k) for k, _ in kvs], dtype=np.int64)
        slots = np.asarray([int(v) for _, v in kvs], dtype=np.int32)
        assert len(set(keys.tolist())) == len(kvs)
        rc = self._L.vlsfr_lru_restore(self._h, keys.ctypes.data, slots.ctypes.data, len(kvs))
        if rc == -2:
            raise AssertionError(self._L.vlsfr_last_error().decode())
        _lib.check(rc, "vlsfr_lru_restore")

    def restore_arrays(self, keys, slots):
        """`restore` for bulk state: int64 keys and int32 slots as arrays, MRU -> LRU, same contract
        (empty LRU, distinct keys, slots a permutation of 0..n-1).  A 10 M-identity pool restores in about a
        second this way; the list-of-tuples form of the reference API needs 10 M Python objects."""
        keys = np.ascontiguousarray(np.asarray(keys, dtype=np.int64))
        slots = np.ascontiguousarray(np.asarray(slots, dtype=np.int32))
        assert keys.ndim == 1 and keys.shape == slots.shape
        assert keys.shape[0] <= self.capacity
        assert self.cur_idx == 0
        rc = self._L.vlsfr_lru_restore(self._h, keys.ctypes.data, slots.ctypes.data, keys.shape[0])
        if rc == -2:
            raise AssertionError(self._L.vlsfr_last_error().decode())
        _lib.check(rc, "vlsfr_lru_restore")

    def state_arrays(self):
        """`state_dict` as (int64 keys, int32 slots) arrays, MRU -> LRU."""
        return self._state_arrays()

    def reset(self):
        """Back to the state of a freshly constructed LRU(capacity) (the reference's clear() keeps cur_idx, so a used
        allocator cannot take restore(); loading a checkpoint into a live run needs this)."""
        self._L.vlsfr_lru_destroy(self._h)
        h = ctypes.c_void_p()
        _lib.check(self._L.vlsfr_lru_create(int(self.capacity), ctypes.byref(h)), "vlsfr_lru_create")
        self._h.value = h.value

    # lru.py:132-141
    def clear(self):
        _lib.check(self._L.vlsfr_lru_clear(self._h), "vlsfr_lru_clear")
