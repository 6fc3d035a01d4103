"""The two helpers of pyvisim/_utils.py that sit on the hot path: the image validator that gates every
extractor call (:34-53) and cosine_similarity (:312-330).  The ~900 lines of plotting / HDF5 / clustering
conveniences of the reference are out of scope."""
from __future__ import annotations

import numpy as np

from ._errors import InvalidImageError

__all__ = ["is_numpy_image", "cosine_similarity"]


def is_numpy_image(image: np.ndarray, pos: int) -> None:
    """2-D arrays must be integer valued (masks); 3-D arrays must be (H, W, 3) within [0, 255]."""
    if len(image.shape) == 2:
        if not np.all(image == image.astype(np.int64)):
            raise InvalidImageError(f"Mask values must be integers. Got min={image.min()} and max={image.max()}.")
    else:
        if image.shape[2] != 3:
            raise InvalidImageError(f"NumPy 3D images must have shape (H, W, 3). Got {image.shape}.")
        if image.min() < 0 or image.max() > 255:
            raise InvalidImageError(
                f"Image values must be in the range [0, 255]. Got min={image.min()} and max={image.max()} "
                f"for position {pos}.")


def _to_numpy(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.cpu().numpy()
    except ImportError:  # torch is optional for this function
        pass
    return np.asarray(x)


def cosine_similarity(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """(N, L), (M, L) -> (N, M) cosine similarities, computed on the MI355X.

    Same contract as pyvisim._utils.cosine_similarity: torch tensors are accepted, 1-D inputs become
    one row, fewer than 2 features raise ValueError, the result is float32 iff both operands are float32
    (else float64), zero rows give zero similarity."""
    from .engine import default_context
    x, y = _to_numpy(x), _to_numpy(y)
    x = x.reshape(1, -1) if len(x.shape) == 1 else x
    y = y.reshape(1, -1) if len(y.shape) == 1 else y
    if x.shape[-1] <= 1 or y.shape[-1] <= 1:
        raise ValueError(f"Cosine similarity requires at least 2 features. Got {x.shape[-1]} features for x "
                         f"and {y.shape[-1]} features for y.")
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    if y.dtype not in (np.float32, np.float64):
        y = y.astype(np.float64)
    return default_context().cosine(x, y)
