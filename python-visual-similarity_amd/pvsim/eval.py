"""Retrieval metrics with the signatures of pyvisim/eval.py:13-145, batched on the MI355X.

The reference loops over queries in Python: encode one query, cosine against the whole database (which
sklearn re-normalises every time), full argsort, slice k.  Here all queries are encoded together, scored
with ONE Q x N similarity GEMM and ranked with the fused top-k kernel; the label bookkeeping is unchanged.
Ranking order is (score descending, index ascending) -- identical to the reference's lists wherever its
own (non-stable) argsort is unambiguous, i.e. on tie-free scores."""
from __future__ import annotations

from typing import Iterable

import numpy as np

from ._utils import cosine_similarity  # re-exported like the reference's `from ._utils import *`
from .engine import default_context

__all__ = ["retrieve_top_k_similar", "top_k_map", "top_k_accuracy"]



def _first_rows(encoder, queries) -> np.ndarray:
    """encoder.encode(query) per query, keeping row 0 (the reference scores `cosine(...)[0]`), stacked."""
    queries = list(queries)
    if queries and all(isinstance(q, np.ndarray) and q.ndim == 3 for q in queries):
        enc = encoder.encode(queries)                       # one batched encode
        return enc.reshape(len(queries), -1) if enc.ndim == 1 else enc
    rows = []
    for q in queries:
        v = encoder.encode(q)
        rows.append(v.reshape(1, -1)[0] if v.ndim == 1 else v[0])
    return np.vstack(rows) if rows else np.zeros((0, 0), np.float32)


def _vectors_and_paths(encoding_map):
    """-> ((N, L) matrix, paths, resident index or None).  A pvsim.index.DeviceIndex already holds the matrix (and a normalised
    copy on the GPU); a plain dict is stacked the way the reference does it (eval.py:28)."""
    from .index import DeviceIndex
    if isinstance(encoding_map, DeviceIndex):
        return encoding_map.matrix, list(encoding_map.keys()), encoding_map
    return np.array(list(encoding_map.values())), list(encoding_map.keys()), None


def _rank(query_vecs: np.ndarray, all_vectors: np.ndarray, k: int | None, ctx=None, resident=None):
    """-> (indices (nq, k') int64, scores (nq, k')) with k' = min(k, N) (k=None: all N)."""
    n = all_vectors.shape[0]
    kk = n if k is None else max(0, min(int(k), n))
    if query_vecs.shape[0] == 0 or kk == 0:
        return np.zeros((query_vecs.shape[0], 0), np.int64), np.zeros((query_vecs.shape[0], 0), np.float32)
    if query_vecs.shape[-1] <= 1 or all_vectors.shape[-1] <= 1:
        raise ValueError(f"Cosine similarity requires at least 2 features. Got {query_vecs.shape[-1]} features "
                         f"for x and {all_vectors.shape[-1]} features for y.")
    if resident is not None:
        return resident.rank(query_vecs, kk)
    ctx = ctx or default_context()
    # the reference's dtype rule (pyvisim/_utils.py:312-330 -> sklearn): float32 scores iff BOTH operands are float32;
    # anything else (Fisher encodings are float64) is scored and ranked in float64
    if query_vecs.dtype == np.float32 and all_vectors.dtype == np.float32:
        return ctx.cosine_topk(np.ascontiguousarray(query_vecs), np.ascontiguousarray(all_vectors), kk)
    return ctx.cosine_topk_f64(query_vecs, all_vectors, kk)


def retrieve_top_k_similar(uploaded_image: np.ndarray, dataset: dict[str, np.ndarray], encoder,
                           k: int = 5) -> list[tuple[str, float]]:
    """[(image_path, similarity)] of the k most similar database entries, best first."""
    all_vectors, all_paths, resident = _vectors_and_paths(dataset)
    query_vector = encoder.encode(uploaded_image)
    if query_vector.ndim == 1:
        query_vector = query_vector.reshape(1, -1)
    idx, val = _rank(query_vector[:1], all_vectors, k, getattr(encoder, "context", None), resident)
    return [(all_paths[i], s) for i, s in zip(idx[0], val[0])]


def top_k_map(images: Iterable[np.ndarray], image_labels: Iterable[int], encoding_map: dict[str, np.ndarray],
              path_labels_dict: dict[str, int], encoder, k: int = None) -> float:
    """Mean average precision; R is counted inside the (possibly truncated) ranked list (eval.py:95)."""
    all_vectors, all_paths, resident = _vectors_and_paths(encoding_map)
    labels = list(image_labels)
    q = _first_rows(encoder, images)
    idx, _ = _rank(q, all_vectors, k, getattr(encoder, "context", None), resident)
    db_labels = [path_labels_dict[p] for p in all_paths]
    aps = []
    for row, true_label in zip(idx, labels):
        relevant_count, precision_sum = 0, 0.0
        for rank, i in enumerate(row, start=1):
            if db_labels[i] == true_label:
                relevant_count += 1
                precision_sum += relevant_count / rank
        aps.append(precision_sum / relevant_count if relevant_count > 0 else 0.0)
    return float(np.mean(aps))


def top_k_accuracy(images: Iterable[np.ndarray], image_labels: Iterable[int], encoding_map: dict[str, np.ndarray],
                   path_labels_dict: dict[str, int], encoder, k: int) -> float:
    """Fraction of queries with at least one same-label entry among their k nearest (eval.py:102-145)."""
    all_vectors, all_paths, resident = _vectors_and_paths(encoding_map)
    images = list(images)
    labels = list(image_labels)
    q = _first_rows(encoder, images)
    idx, _ = _rank(q, all_vectors, k, getattr(encoder, "context", None), resident)
    db_labels = [path_labels_dict[p] for p in all_paths]
    correct = sum(1 for row, true_label in zip(idx, labels) if any(db_labels[i] == true_label for i in row))
    return float(correct / len(images))
