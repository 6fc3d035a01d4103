"""Encoding-map persistence: the reference keeps its index as an in-memory dict {image_path: vector}
(`generate_encoding_map`, pyvisim/encoders/_base_encoder.py:344-359) and only offers generic HDF5 helpers.
Here an index is one plain `.npz` per shard (paths + one (n, L) matrix): no pickles, loadable with
numpy.load(allow_pickle=False), and shaped for the sharded multi-GPU layout (one file per rank)."""
from __future__ import annotations

import os

import numpy as np

__all__ = ["save_encoding_map", "load_encoding_map", "save_shard", "load_shards"]


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
