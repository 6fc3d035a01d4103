"""ctypes wrapper of oracle/_build/libpvsim_oracle.so (pvsim_oracle.c) -- TEST INFRASTRUCTURE ONLY.
Used by tests (second checker) and by bench.py's cpu_baseline leg (the timed CPU port)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libpvsim_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} missing: run `make -C oracle`")
        l = C.CDLL(_PATH)
        l.orc_vlad_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        l.orc_retrieve.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_int]
        l.orc_assign_chain.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        l.orc_cosine_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        _lib = l
    return _lib


def max_threads() -> int:
    return int(lib().orc_max_threads())


def vlad_encode(packed, offsets, centroids, power=1.0, norm_order=2.0, eps=1e-9, threads=0, return_labels=False):
    """packed: (total, D) uint8 raw SIFT (RootSIFT applied inside) or float32 descriptors."""
    is_u8 = packed.dtype == np.uint8
    packed = np.ascontiguousarray(packed, dtype=np.uint8 if is_u8 else np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    n = offsets.shape[0] - 1
    out = np.empty((n, c.shape[0] * c.shape[1]), dtype=np.float32)
    labels = np.empty(packed.shape[0], dtype=np.int32)
    rc = lib().orc_vlad_encode(packed.ctypes.data, int(is_u8), offsets.ctypes.data, n, c.ctypes.data, c.shape[0],
                               c.shape[1], float(power), float(norm_order), float(eps), out.ctypes.data,
                               labels.ctypes.data, int(threads))
    if rc:
        raise MemoryError("oracle allocation failed")
    return (out, labels) if return_labels else out


def retrieve(q, db, k, threads=0):
    q = np.ascontiguousarray(q, dtype=np.float32)
    db = np.ascontiguousarray(db, dtype=np.float32)
    idx = np.empty((q.shape[0], k), dtype=np.int64)
    val = np.empty((q.shape[0], k), dtype=np.float32)
    rc = lib().orc_retrieve(q.ctypes.data, q.shape[0], db.ctypes.data, db.shape[0], q.shape[1], int(k),
                            idx.ctypes.data, val.ctypes.data, int(threads))
    if rc:
        raise MemoryError("oracle allocation failed")
    return idx, val


def cosine_chain(a, b, inv_a, inv_b, pairs):
    """The device's cosine score for the listed (i, j) pairs as its defined fp32 recurrence (see pvsim_oracle.c)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    ia = None if inv_a is None else np.ascontiguousarray(inv_a, dtype=np.float32)
    ib = None if inv_b is None else np.ascontiguousarray(inv_b, dtype=np.float32)
    out = np.empty(pairs.shape[0], dtype=np.float32)
    lib().orc_cosine_chain(a.ctypes.data, b.ctypes.data, a.shape[1], None if ia is None else ia.ctypes.data,
                           None if ib is None else ib.ctypes.data, pairs.ctypes.data, pairs.shape[0], out.ctypes.data)
    return out


def assign_chain(x, centroids):
    """KMeans.predict as the device's exact kernel evaluates it (defined fp32 recurrence, see pvsim_oracle.c)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    labels = np.empty(x.shape[0], dtype=np.int32)
    if lib().orc_assign_chain(x.ctypes.data, x.shape[0], c.ctypes.data, c.shape[0], c.shape[1], labels.ctypes.data):
        raise MemoryError("oracle allocation failed")
    return labels
