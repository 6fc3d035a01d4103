"""Thin object layer over the C-ABI: one Context = one MI355X + one HIP stream; device tables behind
handles.  Everything here marshals pointers and sizes -- no arithmetic happens in Python."""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

from . import _ffi
from ._ffi import DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT, check, norm_params, ptr

__all__ = ["Context", "default_context", "pack_descriptors", "DESC_F32", "DESC_F32_ROOTSIFT", "DESC_U8_ROOTSIFT"]


def pack_descriptors(desc_list, dim: int, dtype=np.float32):
    """list of (n_i, D) arrays -> (packed (sum n_i, D) C-contiguous, offsets int64 (N+1,))."""
    counts = [0 if d is None else int(d.shape[0]) for d in desc_list]
    offsets = np.zeros(len(desc_list) + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    packed = np.empty((int(offsets[-1]), dim), dtype=dtype)
    for d, o in zip(desc_list, offsets[:-1]):
        if d is not None and d.shape[0]:
            if d.shape[1] != dim:
                raise RuntimeError(f"descriptor dimension {d.shape[1]} does not match the model input dimension {dim}")
            packed[o:o + d.shape[0]] = d
    return packed, offsets


class _Handle:
    _destroy = None

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.handle = ctx, handle

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            getattr(_ffi.lib(), self._destroy)(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Codebook(_Handle):
    _destroy = "pvs_codebook_destroy"
    K = D = 0


class GMM(_Handle):
    _destroy = "pvs_gmm_destroy"
    K = D = 0


class PCATable(_Handle):
    _destroy = "pvs_pca_destroy"
    C = Din = 0


class DeviceBuffer:
    """A block of device memory owned by a Context (pvs_malloc / pvs_free); .ptr is the raw device address."""

    def __init__(self, ctx: "Context", nbytes: int):
        self._ctx, self.nbytes, self.ptr = ctx, int(nbytes), 0
        self._cap = max(int(nbytes), 16)
        hit = ctx._take_cached(self._cap)
        if hit is not None:                      # a block this context released earlier: no hipMalloc / hipFree on the hot path
            self.ptr, self._cap = hit
            return
        p = C.c_void_p()
        check(_ffi.lib().pvs_malloc(ctx.handle, self._cap, C.byref(p)))
        self.ptr = int(p.value)

    @classmethod
    def view(cls, ctx: "Context", dptr: int, nbytes: int) -> "DeviceBuffer":
        """Non-owning view of device memory allocated elsewhere (e.g. torch.Tensor.data_ptr()); free() is a no-op."""
        b = cls.__new__(cls)
        b._ctx, b.nbytes, b.ptr, b._borrowed = ctx, int(nbytes), int(dptr), True
        return b

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        if offset + a.nbytes > self.nbytes:
            raise ValueError("upload past the end of the device buffer")
        check(_ffi.lib().pvs_memcpy_h2d(self._ctx.handle, C.c_void_p(self.ptr + offset), ptr(a), a.nbytes))
        return self

    def download(self, shape, dtype, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if offset + out.nbytes > self.nbytes:
            raise ValueError("download past the end of the device buffer")
        check(_ffi.lib().pvs_memcpy_d2h(self._ctx.handle, ptr(out), C.c_void_p(self.ptr + offset), out.nbytes))
        return out

    def fill_bytes(self, value: int):
        check(_ffi.lib().pvs_memset(self._ctx.handle, C.c_void_p(self.ptr), int(value), self.nbytes))
        return self

    def free(self):
        if self.ptr and not getattr(self, "_borrowed", False) and self._ctx.handle is not None:
            if not self._ctx._give_cached(self.ptr, getattr(self, "_cap", self.nbytes)):
                _ffi.lib().pvs_free(self._ctx.handle, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One device + one stream.  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream, so that work interleaves with torch in order."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.handle = None
        h = C.c_void_p()
        check(_ffi.lib().pvs_init(int(device), C.c_void_p(stream), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._cache = []          # released DeviceBuffers kept for reuse: [(capacity, ptr)], bounded (training loops allocate the same sizes over and over; hipFree synchronises the device)

    _CACHE_BLOCKS, _CACHE_BYTES = 24, 2 << 30

    def _take_cached(self, nbytes):
        best = None
        for i, (cap, p) in enumerate(self._cache):
            if nbytes <= cap <= 2 * nbytes + 4096 and (best is None or cap < self._cache[best][0]):
                best = i
        if best is None:
            return None
        cap, p = self._cache.pop(best)
        return p, cap

    def _give_cached(self, p, cap):
        if len(self._cache) >= self._CACHE_BLOCKS or cap + sum(c for c, _ in self._cache) > self._CACHE_BYTES:
            return False
        self._cache.append((int(cap), int(p)))
        return True

    def trim(self):
        """Return the cached device blocks to the driver."""
        while self._cache:
            _, p = self._cache.pop()
            if self.handle is not None:
                _ffi.lib().pvs_free(self.handle, C.c_void_p(p))

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if self.handle is not None:
            self.trim()
            _ffi.lib().pvs_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        check(_ffi.lib().pvs_sync(self.handle))

    @property
    def stream(self) -> int:
        return int(_ffi.lib().pvs_stream(self.handle) or 0)

    def set_option(self, option: int, value: int) -> None:
        """pvs_set_option: pins one of several implementations that must agree bit for bit (tests, benchmarks)."""
        check(_ffi.lib().pvs_set_option(self.handle, int(option), int(value)))

    def get_option(self, option: int) -> int:
        v = C.c_int()
        check(_ffi.lib().pvs_get_option(self.handle, int(option), C.byref(v)))
        return int(v.value)

    def option(self, option: int, value: int):
        """Context manager: `with ctx.option(OPT_ASSIGN_PREFILTER, 0): ...` restores the previous value on exit."""
        ctx = self

        class _Scoped:
            def __enter__(self_inner):
                self_inner.old = ctx.get_option(option)
                ctx.set_option(option, value)

            def __exit__(self_inner, *exc):
                ctx.set_option(option, self_inner.old)

        return _Scoped()

    def fill_dev(self, d_ptr: int, n_elems: int, dtype: str, value) -> None:
        """4- or 8-byte pattern fill on this context's stream (dtype: float32 / int32 / int64 / float64)."""
        a = np.array([value], dtype=dtype)
        if a.itemsize not in (4, 8):
            raise ValueError("fill_dev: 4- or 8-byte element types only")
        pattern = int(a.view(np.uint32 if a.itemsize == 4 else np.uint64)[0])
        check(_ffi.lib().pvs_fill_dev(self.handle, ptr(d_ptr), int(n_elems), a.itemsize, C.c_uint64(pattern)))

    def wait_for(self, other: "Context") -> None:
        """Work queued on THIS context after the call waits for everything `other` has queued so far (no host sync)."""
        check(_ffi.lib().pvs_stream_wait(self.handle, other.handle))

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        check(_ffi.lib().pvs_device_name(self.handle, buf, 256))
        return buf.value.decode()

    # ------------------------------------------------------------------ tables
    def codebook(self, centroids) -> Codebook:
        c = np.ascontiguousarray(centroids, dtype=np.float32)
        if c.ndim != 2:
            raise ValueError("centroids must be (K, D)")
        h = C.c_void_p()
        check(_ffi.lib().pvs_codebook_create(self.handle, ptr(c), c.shape[0], c.shape[1], C.byref(h)))
        cb = Codebook(self, h)
        cb.K, cb.D = c.shape
        return cb

    def gmm(self, weights, means, covariances) -> GMM:
        w = np.ascontiguousarray(weights, dtype=np.float64)
        m = np.ascontiguousarray(means, dtype=np.float64)
        v = np.ascontiguousarray(covariances, dtype=np.float64)
        if m.ndim != 2 or v.shape != m.shape or w.shape != (m.shape[0],):
            raise ValueError("GMM tables must be weights (K,), means (K, D), diagonal covariances (K, D)")
        h = C.c_void_p()
        check(_ffi.lib().pvs_gmm_create(self.handle, ptr(w), ptr(m), ptr(v), m.shape[0], m.shape[1], C.byref(h)))
        g = GMM(self, h)
        g.K, g.D = m.shape
        return g

    def pca(self, components, mean) -> PCATable:
        comp = np.ascontiguousarray(components, dtype=np.float32)
        mu = np.ascontiguousarray(mean, dtype=np.float32).reshape(-1)
        if comp.ndim != 2 or mu.shape[0] != comp.shape[1]:
            raise ValueError("PCA tables must be components (C, Din), mean (Din,)")
        h = C.c_void_p()
        check(_ffi.lib().pvs_pca_create(self.handle, ptr(comp), ptr(mu), comp.shape[0], comp.shape[1], C.byref(h)))
        p = PCATable(self, h)
        p.C, p.Din = comp.shape
        return p

    # ------------------------------------------------------------------ encoders (host arrays)
    @staticmethod
    def _check_desc(packed, kind, dim):
        want = np.uint8 if kind == DESC_U8_ROOTSIFT else np.float32
        packed = np.ascontiguousarray(packed, dtype=want)
        if packed.ndim != 2 or packed.shape[1] != dim:
            raise RuntimeError(f"descriptors must be (n, {dim}), got {packed.shape}")
        return packed

    def vlad_encode(self, cb: Codebook, packed, offsets, kind=DESC_F32, power=1.0, norm_order=2, epsilon=1e-9,
                    pca: PCATable | None = None, return_labels=False):
        d_in = pca.Din if pca is not None else cb.D
        packed = self._check_desc(packed, kind, d_in)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = offsets.shape[0] - 1
        if offsets[-1] != packed.shape[0]:
            raise ValueError("offsets[-1] must equal the number of descriptor rows")
        out = np.empty((n, cb.K * cb.D), dtype=np.float32)
        labels = np.empty(packed.shape[0], dtype=np.int32) if return_labels else None
        prm = norm_params(power, norm_order, epsilon)
        check(_ffi.lib().pvs_vlad_encode(self.handle, cb.handle, pca.handle if pca else None, ptr(packed), kind,
                                         ptr(offsets), n, C.byref(prm), ptr(out), ptr(labels)))
        return (out, labels) if return_labels else out

    def fisher_encode(self, g: GMM, packed, offsets, kind=DESC_F32, power=0.5, norm_order=2, epsilon=1e-9,
                      pca: PCATable | None = None):
        d_in = pca.Din if pca is not None else g.D
        packed = self._check_desc(packed, kind, d_in)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = offsets.shape[0] - 1
        if offsets[-1] != packed.shape[0]:
            raise ValueError("offsets[-1] must equal the number of descriptor rows")
        out = np.empty((n, g.K + 2 * g.K * g.D), dtype=np.float64)
        prm = norm_params(power, norm_order, epsilon)
        # the device entry takes at most 65535 images per launch: batch here
        step = 32768
        for s in range(0, n, step):
            e = min(n, s + step)
            sub_off = offsets[s:e + 1] - offsets[s]
            sub = packed[offsets[s]:offsets[e]]
            check(_ffi.lib().pvs_fisher_encode(self.handle, g.handle, pca.handle if pca else None, ptr(sub), kind,
                                               ptr(np.ascontiguousarray(sub_off)), e - s, C.byref(prm), ptr(out[s:e])))
        return out

    # ------------------------------------------------------------------ similarity (host arrays)
    def cosine(self, a, b):
        a, b = np.asarray(a), np.asarray(b)
        f64 = not (a.dtype == np.float32 and b.dtype == np.float32)
        dt = np.float64 if f64 else np.float32
        a = np.ascontiguousarray(a, dtype=dt)
        b = np.ascontiguousarray(b, dtype=dt)
        if a.shape[1] != b.shape[1]:
            raise ValueError(f"Incompatible dimension for X and Y matrices: X.shape[1] == {a.shape[1]} "
                             f"while Y.shape[1] == {b.shape[1]}")
        out = np.empty((a.shape[0], b.shape[0]), dtype=dt)
        same = a is b or (a.ctypes.data == b.ctypes.data and a.shape == b.shape)
        check(_ffi.lib().pvs_cosine(self.handle, ptr(a), a.shape[0], ptr(a if same else b), b.shape[0], a.shape[1],
                                    int(f64), ptr(out)))
        return out

    def cosine_topk(self, q, db, k: int):
        q = np.ascontiguousarray(q, dtype=np.float32)
        db = np.ascontiguousarray(db, dtype=np.float32)
        if q.shape[1] != db.shape[1]:
            raise ValueError("query and database dimensions differ")
        idx = np.empty((q.shape[0], k), dtype=np.int64)
        val = np.empty((q.shape[0], k), dtype=np.float32)
        same = q.ctypes.data == db.ctypes.data and q.shape == db.shape
        check(_ffi.lib().pvs_cosine_topk(self.handle, ptr(q), q.shape[0], ptr(q if same else db), db.shape[0],
                                         q.shape[1], int(k), ptr(idx), ptr(val)))
        return idx, val

    def cosine_topk_f64(self, q, db, k: int):
        """float64 scores and ranking (the reference's dtype whenever an operand is not float32, e.g. Fisher encodings)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        db = np.ascontiguousarray(db, dtype=np.float64)
        if q.shape[1] != db.shape[1]:
            raise ValueError("query and database dimensions differ")
        idx = np.empty((q.shape[0], k), dtype=np.int64)
        val = np.empty((q.shape[0], k), dtype=np.float64)
        check(_ffi.lib().pvs_cosine_topk_f64(self.handle, ptr(q), q.shape[0], ptr(db), db.shape[0], q.shape[1], int(k),
                                             ptr(idx), ptr(val)))
        return idx, val

    # ------------------------------------------------------------------ device-pointer forms
    # All pointer arguments are raw device addresses (int), e.g. torch.Tensor.data_ptr().
    def vlad_encode_dev(self, cb, d_desc, kind, d_offsets, n_images, total_desc, d_out, power=1.0, norm_order=2,
                        epsilon=1e-9, pca=None, d_labels=None, d_inv_norm=None):
        prm = norm_params(power, norm_order, epsilon)
        check(_ffi.lib().pvs_vlad_encode_dev(self.handle, cb.handle, pca.handle if pca else None, ptr(d_desc), kind,
                                             ptr(d_offsets), n_images, total_desc, C.byref(prm), ptr(d_out),
                                             ptr(d_labels), ptr(d_inv_norm)))

    def fisher_encode_dev(self, g, d_desc, kind, d_offsets, n_images, total_desc, d_out, out_f64, power=0.5,
                          norm_order=2, epsilon=1e-9, pca=None):
        prm = norm_params(power, norm_order, epsilon)
        check(_ffi.lib().pvs_fisher_encode_dev(self.handle, g.handle, pca.handle if pca else None, ptr(d_desc), kind,
                                               ptr(d_offsets), n_images, total_desc, C.byref(prm), ptr(d_out),
                                               int(out_f64)))

    def kmeans_predict_dev(self, cb, d_desc, kind, total_desc, d_labels):
        check(_ffi.lib().pvs_kmeans_predict_dev(self.handle, cb.handle, ptr(d_desc), kind, total_desc, ptr(d_labels)))

    def gmm_predict_proba_dev(self, g, d_desc, kind, total_desc, d_resp):
        check(_ffi.lib().pvs_gmm_predict_proba_dev(self.handle, g.handle, ptr(d_desc), kind, total_desc, ptr(d_resp)))

    def pca_transform_dev(self, p, d_desc, kind, total_desc, d_out):
        check(_ffi.lib().pvs_pca_transform_dev(self.handle, p.handle, ptr(d_desc), kind, total_desc, ptr(d_out)))

    def row_inv_norms_dev(self, d_x, rows, L, d_inv):
        check(_ffi.lib().pvs_row_inv_norms_dev(self.handle, ptr(d_x), rows, L, ptr(d_inv)))

    def cosine_dev(self, d_a, M, d_b, N, L, d_inva, d_invb, d_out, ldo):
        check(_ffi.lib().pvs_cosine_dev(self.handle, ptr(d_a), M, ptr(d_b), N, L, ptr(d_inva), ptr(d_invb), ptr(d_out), ldo))

    def cosine_dual_dev(self, d_a, M, d_b, N, L, d_inva, d_invb, d_out, ldo, d_out_t, ldt):
        check(_ffi.lib().pvs_cosine_dual_dev(self.handle, ptr(d_a), M, ptr(d_b), N, L, ptr(d_inva), ptr(d_invb), ptr(d_out),
                                             ldo, ptr(d_out_t), ldt))

    def topk_dev(self, d_scores, nq, ncols, ld, k, col_offset, merge, d_idx, d_val):
        check(_ffi.lib().pvs_topk_dev(self.handle, ptr(d_scores), nq, ncols, ld, k, col_offset, int(merge), ptr(d_idx),
                                      ptr(d_val)))

    def cosine_topk_dev(self, d_q, nq, d_db, N, L, d_invq, d_invdb, k, col_offset, merge, d_idx, d_val):
        check(_ffi.lib().pvs_cosine_topk_dev(self.handle, ptr(d_q), nq, ptr(d_db), N, L, ptr(d_invq), ptr(d_invdb), k,
                                             col_offset, int(merge), ptr(d_idx), ptr(d_val)))

    def cosine_topk_filtered_dev(self, d_q, nq, d_db, N, L, d_invq, d_invdb, k, d_idx, d_val):
        """Same lists as cosine_topk_dev (bit-identical), via the fp16 prefilter + exact re-scoring.
        -> dict(filtered, redone_exact, candidates, slots)"""
        st = (C.c_int64 * 4)()
        check(_ffi.lib().pvs_cosine_topk_filtered_dev(self.handle, ptr(d_q), nq, ptr(d_db), N, L, ptr(d_invq), ptr(d_invdb), k,
                                                      ptr(d_idx), ptr(d_val), st))
        return {"filtered": bool(st[0]), "redone_exact": int(st[1]), "candidates": int(st[2]), "slots": int(st[3])}

    def f32_to_f16_dev(self, d_src, n, d_dst):
        check(_ffi.lib().pvs_f32_to_f16_dev(self.handle, ptr(d_src), n, ptr(d_dst)))

    def cosine_f16_dev(self, d_a, M, d_b, N, L, d_inva, d_invb, d_out, ldo):
        check(_ffi.lib().pvs_cosine_f16_dev(self.handle, ptr(d_a), M, ptr(d_b), N, L, ptr(d_inva), ptr(d_invb), ptr(d_out), ldo))

    def cosine_topk_f16_dev(self, d_q, nq, d_db, N, L, d_invq, d_invdb, k, col_offset, merge, d_idx, d_val):
        check(_ffi.lib().pvs_cosine_topk_f16_dev(self.handle, ptr(d_q), nq, ptr(d_db), N, L, ptr(d_invq), ptr(d_invdb), k,
                                                 col_offset, int(merge), ptr(d_idx), ptr(d_val)))

    def row_inv_norms_f64_dev(self, d_x, rows, L, d_inv):
        check(_ffi.lib().pvs_row_inv_norms_f64_dev(self.handle, ptr(d_x), rows, L, ptr(d_inv)))

    def cosine_f64_dev(self, d_a, M, d_b, N, L, d_inva, d_invb, d_out, ldo):
        """float64 operands and scores on the f64 matrix pipe (the reference's dtype for Fisher encodings)"""
        check(_ffi.lib().pvs_cosine_f64_dev(self.handle, ptr(d_a), M, ptr(d_b), N, L, ptr(d_inva), ptr(d_invb), ptr(d_out), ldo))

    def cosine_topk_f64_dev(self, d_q, nq, d_db, N, L, d_invq, d_invdb, k, d_idx, d_val):
        """float64 scores + ranking, everything resident: d_idx int64 [nq][k], d_val float64 [nq][k]"""
        check(_ffi.lib().pvs_cosine_topk_f64_dev(self.handle, ptr(d_q), nq, ptr(d_db), N, L, ptr(d_invq), ptr(d_invdb), int(k),
                                                 ptr(d_idx), ptr(d_val)))

    def topk_merge_dev(self, d_idx_lists, d_val_lists, n_lists, nq, k, d_idx, d_val):
        check(_ffi.lib().pvs_topk_merge_dev(self.handle, ptr(d_idx_lists), ptr(d_val_lists), n_lists, nq, k, ptr(d_idx),
                                            ptr(d_val)))

    # ------------------------------------------------------------------ vocabulary training (one device pass each)
    def buffer(self, nbytes: int) -> "DeviceBuffer":
        return DeviceBuffer(self, nbytes)

    def materialise_dev(self, d_desc, kind, D, total_desc, d_out):
        check(_ffi.lib().pvs_materialise_dev(self.handle, ptr(d_desc), kind, D, total_desc, ptr(d_out)))

    def kmeans_step_dev(self, cb, d_x, total_desc, d_labels, d_prev_labels=None, d_sqdist=None):
        """-> (residual sums (K, D) f64, counts (K,) f64, inertia, changed labels)"""
        K, D = cb.K, cb.D
        st = np.empty(K * D + K + 2, dtype=np.float64)
        check(_ffi.lib().pvs_kmeans_step_dev(self.handle, cb.handle, ptr(d_x), total_desc, ptr(d_labels), ptr(d_prev_labels),
                                             ptr(st), ptr(d_sqdist)))
        return st[:K * D].reshape(K, D), st[K * D:K * D + K], float(st[K * D + K]), int(st[K * D + K + 1])

    def gmm_em_step_dev(self, g, d_x, total_desc):
        """-> (s0 (K,), s1 (K, D), s2 (K, D), sum_i log p(x_i)), all fp64"""
        K, D = g.K, g.D
        st = np.empty(K + 2 * K * D + 1, dtype=np.float64)
        check(_ffi.lib().pvs_gmm_em_step_dev(self.handle, g.handle, ptr(d_x), total_desc, ptr(st)))
        s12 = st[K:K + 2 * K * D].reshape(K, 2, D)
        return st[:K], s12[:, 0, :], s12[:, 1, :], float(st[-1])

    def gram_dev(self, d_x, D, total_desc):
        """-> (sum_i x_i (D,), sum_i x_i x_i^T (D, D)) in fp64"""
        out = np.empty(D + D * D, dtype=np.float64)
        check(_ffi.lib().pvs_gram_dev(self.handle, ptr(d_x), D, total_desc, ptr(out)))
        return out[:D], out[D:].reshape(D, D)

    def label_sums_dev(self, d_x, D, total_desc, d_labels, K, square=False):
        out = np.empty((K, D), dtype=np.float64)
        check(_ffi.lib().pvs_label_sums_dev(self.handle, ptr(d_x), D, total_desc, ptr(d_labels), K, int(square), ptr(out)))
        return out

    def seed_distances_dev(self, d_x, D, total_desc, cand, d_mind, d_dist, n_cand=None):
        """cand: host (n_cand, D) array, or a raw device pointer (int) together with n_cand"""
        on_dev = not isinstance(cand, np.ndarray)
        if not on_dev:
            cand = np.ascontiguousarray(cand, dtype=np.float32).reshape(-1, D)
            n_cand = cand.shape[0]
        pot = np.empty(n_cand, dtype=np.float64)
        check(_ffi.lib().pvs_seed_distances_dev(self.handle, ptr(d_x), D, total_desc, ptr(cand), n_cand, ptr(d_mind),
                                                ptr(d_dist), ptr(pot), int(on_dev)))
        return pot

    def seed_pick_dev(self, d_x, D, total_desc, d_mind, blocks, base, target, d_cand):
        blocks = np.ascontiguousarray(blocks, dtype=np.int64)
        base = np.ascontiguousarray(base, dtype=np.float64)
        target = np.ascontiguousarray(target, dtype=np.float64)
        idx = np.empty(len(blocks), dtype=np.int64)
        check(_ffi.lib().pvs_seed_pick_dev(self.handle, ptr(d_x), D, total_desc, ptr(d_mind), ptr(blocks), ptr(base), ptr(target),
                                           len(blocks), ptr(d_cand), ptr(idx)))
        return idx

    def min_update_dev(self, d_mind, d_dist, total_desc):
        bs = np.empty((total_desc + 4095) // 4096, dtype=np.float64)
        check(_ffi.lib().pvs_min_update_dev(self.handle, ptr(d_mind), ptr(d_dist), total_desc, ptr(bs)))
        return bs

    def kmeanspp_run_dev(self, d_x, D, total_desc, n_clusters, trials, uniform, first_index):
        """pvs_kmeanspp_run_dev: the greedy k-means++ run on the device (trials <= 8) -> indices (n_clusters,) int64."""
        u = np.ascontiguousarray(uniform, dtype=np.float64).reshape(-1)
        if u.size < max(n_clusters - 1, 0) * trials:
            raise ValueError("need (n_clusters - 1) x trials uniform numbers")
        idx = np.full(n_clusters, -1, dtype=np.int64)
        idx[0] = first_index
        if u.size == 0:
            u = np.zeros(1)
        check(_ffi.lib().pvs_kmeanspp_run_dev(self.handle, ptr(d_x), D, total_desc, n_clusters, trials, ptr(u), ptr(idx)))
        return idx

    def fused_profile(self, enable=True, raw=False):
        """pvs_fused_profile: (re)start or stop the stamped diagnostic builds of the VLAD encode kernels -> the 16 counters so far
        (fused path: the names below; two-kernel path with raw=True: the prefilter's phase cycles, see tests/tools/assign_profile.py)."""
        out = np.zeros(16, dtype=np.int64)
        check(_ffi.lib().pvs_fused_profile(self.handle, int(enable), ptr(out)))
        if raw:
            return out
        names = ("P0", "A", "reduce", "reevaluate", "K2", "epilogue", "image_switch", "_", "stages", "stages_reevaluated", "entries", "rows_unsettled")
        return {n: int(v) for n, v in zip(names, out) if n != "_"}

    # ------------------------------------------------------------------ timers
    def timers_enable(self, on=True):
        check(_ffi.lib().pvs_timers_enable(self.handle, int(on)))

    def timers_reset(self):
        check(_ffi.lib().pvs_timers_reset(self.handle))

    def timers(self) -> dict:
        out = {}
        for i, name in enumerate(_ffi.TIMER_NAMES):
            ms, cnt = C.c_double(), C.c_int64()
            check(_ffi.lib().pvs_timers_read(self.handle, i, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out


_default = None
_default_lock = threading.Lock()


def default_context() -> Context:
    """Process-wide context on device LOCAL_RANK (0 if unset); created on first use."""
    global _default
    if _default is None or _default.handle is None:
        with _default_lock:
            if _default is None or _default.handle is None:
                _default = Context(int(os.environ.get("PVSIM_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default
