"""Encoding-map persistence: the reference keeps its index as an in-memory dict {image_path: vector}
(`generate_encoding_map`, pyvisim/encoders/_base_encoder.py:344-359) and only offers generic HDF5 helpers.
Here an index is one plain `.npz` per shard (paths + one (n, L) matrix): no pickles, loadable with
numpy.load(allow_pickle=False), and shaped for the sharded multi-GPU layout (one file per rank)."""
from __future__ import annotations

import os
from collections.abc import Mapping

import numpy as np

__all__ = ["save_encoding_map", "load_encoding_map", "save_shard", "load_shards", "DeviceIndex"]


def save_encoding_map(path: str, encoding_map: dict) -> None:
    """{path: (L,) vector} -> <path>.npz, preserving insertion order (= database index order, eval.py:28)."""
    keys = list(encoding_map.keys())
    mat = np.vstack([np.asarray(encoding_map[k]).reshape(1, -1) for k in keys]) if keys else np.zeros((0, 0), np.float32)
    np.savez(path, paths=np.array(keys, dtype=np.str_), vectors=mat)


def load_encoding_map(path: str) -> dict:
    with np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False) as z:
        return dict(zip([str(p) for p in z["paths"]], z["vectors"]))


def save_shard(directory: str, rank: int, world: int, first_index: int, vectors: np.ndarray, paths=None) -> str:
    """One rank's contiguous block of the corpus (images [first_index, first_index + n))."""
    os.makedirs(directory, exist_ok=True)
    fn = os.path.join(directory, f"shard_{rank:04d}_of_{world:04d}.npz")
    np.savez(fn, first_index=np.int64(first_index), vectors=np.ascontiguousarray(vectors),
             paths=np.array(list(paths) if paths is not None else [], dtype=np.str_))
    return fn


def load_shards(directory: str):
    """-> (vectors (N, L) in global index order, paths list) from every shard file of `directory`."""
    files = sorted(f for f in os.listdir(directory) if f.startswith("shard_") and f.endswith(".npz"))
    parts = []
    for f in files:
        with np.load(os.path.join(directory, f), allow_pickle=False) as z:
            parts.append((int(z["first_index"]), z["vectors"], [str(p) for p in z["paths"]]))
    parts.sort(key=lambda p: p[0])
    vecs = np.vstack([p[1] for p in parts]) if parts else np.zeros((0, 0), np.float32)
    return vecs, [q for p in parts for q in p[2]]


class DeviceIndex(Mapping):
    """An encoding map {image_path: vector} whose vectors ALSO live on the GPU, uploaded and normalised once.

    The reference's retrieval functions take the index as a dict and rebuild the (N, L) matrix from it on every call
    (`np.array(list(dataset.values()))`, pyvisim/eval.py:28,65,121) -- for one query image against a resident database that copy
    and its upload are the whole cost.  A DeviceIndex is a read-only Mapping with the dict's keys, order and rows, so it can be
    passed wherever `eval.retrieve_top_k_similar` / `top_k_map` / `top_k_accuracy` take `dataset` / `encoding_map`; they then
    rank against the resident copy (pvs_cosine_topk_dev / pvs_cosine_topk_filtered_dev / pvs_cosine_topk_f64_dev: the same lists
    and scores, bit for bit, as with the dict).  dtype rule of the reference (pyvisim/_utils.py:312-330): float32 scores iff the
    database AND the queries are float32, float64 otherwise."""

    def __init__(self, encoding_map, ctx=None):
        from .engine import default_context
        self.ctx = ctx or default_context()
        self._paths = list(encoding_map.keys())
        mat = np.array(list(encoding_map.values()))                      # as the reference builds it (eval.py:28)
        if mat.ndim != 2:
            raise ValueError("DeviceIndex needs one vector of the same length per entry")
        self._host = np.ascontiguousarray(mat, dtype=np.float32 if mat.dtype == np.float32 else np.float64)
        self._pos = {p: i for i, p in enumerate(self._paths)}
        n, L = self._host.shape
        self._db = self.ctx.buffer(max(self._host.nbytes, 16))
        self._inv = self.ctx.buffer(max(n, 1) * self._host.itemsize)
        if n and L:
            self._db.upload(self._host)
            if self._host.dtype == np.float32:
                self.ctx.row_inv_norms_dev(self._db.ptr, n, L, self._inv.ptr)
            else:
                self.ctx.row_inv_norms_f64_dev(self._db.ptr, n, L, self._inv.ptr)

    # ---- Mapping: the dict's view of the same rows
    def __getitem__(self, path):
        return self._host[self._pos[path]]

    def __iter__(self):
        return iter(self._paths)

    def __len__(self):
        return len(self._paths)

    @property
    def matrix(self) -> np.ndarray:
        """(N, L) host copy in index order (= np.array(list(d.values())))."""
        return self._host

    def rank(self, query_vecs: np.ndarray, k: int):
        """-> (idx (nq, k) int64, val (nq, k)) of the queries against the resident database; 1 <= k <= N."""
        q = np.asarray(query_vecs)
        n, L = self._host.shape
        if q.ndim != 2 or q.shape[1] != L:
            raise ValueError("query and database dimensions differ")
        nq = q.shape[0]
        f32 = self._host.dtype == np.float32 and q.dtype == np.float32
        if not f32 and self._host.dtype == np.float32:                   # mixed dtypes: float64 scores from the host copies
            return self.ctx.cosine_topk_f64(q, self._host, int(k))
        q = np.ascontiguousarray(q, dtype=self._host.dtype)
        isz = q.itemsize
        d_q = self.ctx.buffer(max(q.nbytes, 16)).upload(q)
        d_invq = self.ctx.buffer(max(nq, 1) * isz)
        d_idx = self.ctx.buffer(nq * k * 8)
        d_val = self.ctx.buffer(nq * k * isz)
        try:
            if f32:
                self.ctx.row_inv_norms_dev(d_q.ptr, nq, L, d_invq.ptr)
                if nq >= 512:     # as the host entry point: the filtered retrieval gives the same lists faster, and declines what does not qualify
                    self.ctx.cosine_topk_filtered_dev(d_q.ptr, nq, self._db.ptr, n, L, d_invq.ptr, self._inv.ptr, int(k), d_idx.ptr, d_val.ptr)
                else:
                    self.ctx.cosine_topk_dev(d_q.ptr, nq, self._db.ptr, n, L, d_invq.ptr, self._inv.ptr, int(k), 0, False, d_idx.ptr, d_val.ptr)
            else:
                self.ctx.row_inv_norms_f64_dev(d_q.ptr, nq, L, d_invq.ptr)
                self.ctx.cosine_topk_f64_dev(d_q.ptr, nq, self._db.ptr, n, L, d_invq.ptr, self._inv.ptr, int(k), d_idx.ptr, d_val.ptr)
            return d_idx.download((nq, k), np.int64), d_val.download((nq, k), q.dtype)
        finally:
            for b in (d_q, d_invq, d_idx, d_val):
                b.free()

    def close(self):
        for b in (self._db, self._inv):
            b.free()
