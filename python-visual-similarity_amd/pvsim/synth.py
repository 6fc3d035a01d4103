"""Seeded synthetic inputs shared by the golden generator, the tests and bench.py.

Nothing here comes from the reference: OpenCV (SIFT) and the Oxford-102 images
are absent offline, so the hot path is driven at the
*descriptor* level with SIFT-like integer descriptors (0..255, as OpenCV's
`detectAndCompute` returns them) that the RootSIFT transform then normalises.
"""
from __future__ import annotations

import numpy as np

SIFT_DIM = 128


def sift_prototypes(seed: int = 7, n_proto: int = 96, dim: int = SIFT_DIM) -> np.ndarray:
    """Sparse positive prototype histograms (gradient-orientation-bin like)."""
    rng = np.random.default_rng(seed)
    proto = rng.gamma(shape=0.45, scale=40.0, size=(n_proto, dim))
    # a third of the bins of every prototype are (almost) always empty, like real SIFT
    mask = rng.random((n_proto, dim)) < 0.33
    proto[mask] = 0.0
    return proto.astype(np.float64)


def sift_like(n: int, rng: np.random.Generator, proto: np.ndarray | None = None) -> np.ndarray:
    """(n, 128) float32, integer valued in [0, 255] -- stands in for cv2.SIFT output."""
    if proto is None:
        proto = sift_prototypes()
    dim = proto.shape[1]
    if n == 0:
        return np.zeros((0, dim), dtype=np.float32)
    z = rng.integers(0, proto.shape[0], size=n)
    x = proto[z] * rng.lognormal(0.0, 0.35, size=(n, dim)) + rng.gamma(0.3, 4.0, size=(n, dim))
    x *= 512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    x = np.minimum(x, 255.0)
    return np.rint(x).astype(np.float32)


def rootsift(desc: np.ndarray) -> np.ndarray:
    """RootSIFT tail, same arithmetic as the reference extractor
    (pyvisim/features/_features.py:112-114): fp32, d /= sum+1e-7, sqrt."""
    d = np.array(desc, dtype=np.float32, copy=True)
    d /= (d.sum(axis=1, keepdims=True) + np.float32(1e-7))
    return np.sqrt(d)


def ragged_counts(n_images: int, seed: int, mean_n: float = 1257.0, sigma: float = 0.5,
                  lo: int = 50, hi: int = 4000) -> np.ndarray:
    """Descriptor counts per image: round(LogNormal(ln mean_n, sigma)) clipped."""
    rng = np.random.default_rng(seed)
    n = np.rint(rng.lognormal(np.log(mean_n), sigma, size=n_images)).astype(np.int64)
    return np.clip(n, lo, hi)


def deep_like(n: int, dim: int, rng: np.random.Generator, centers: np.ndarray) -> np.ndarray:
    """(n, dim) float32 'deep feature' rows: a random mixture centre + gaussian noise."""
    z = rng.integers(0, centers.shape[0], size=n)
    return (centers[z] + rng.normal(0.0, 0.7, size=(n, dim))).astype(np.float32)
