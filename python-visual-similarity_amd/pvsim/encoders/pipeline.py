"""Pipeline -- concatenation of several encoders' flattened outputs (pyvisim/encoders/pipeline.py:15-127)."""
from __future__ import annotations

import logging
from itertools import tee
from typing import Callable, Iterable

import numpy as np

from .._base_classes import SimilarityMetric
from .._utils import cosine_similarity
from ._base_encoder import ImageEncoderBase, check_desired_output, _is_torch_tensor


class Pipeline(SimilarityMetric):
    """:param encoders: list of ImageEncoderBase instances; :param similarity_func: (N, L), (M, L) -> (N, M)"""
    _logger = logging.getLogger("Pipeline")

    def __init__(self, encoders: list[ImageEncoderBase],
                 similarity_func: Callable[[np.ndarray, np.ndarray], float] = cosine_similarity):
        for encoder in encoders:
            if not isinstance(encoder, ImageEncoderBase):
                raise ValueError(f"Pipeline only accepts instances of ImageEncoderBase, not {type(encoder)}")
        self.encoders = encoders
        self._similarity_func = similarity_func

    def encode(self, images: Iterable[np.ndarray] | np.ndarray) -> np.ndarray:
        if _is_torch_tensor(images):
            raise RuntimeError("Torch images are not supported yet.")
        if isinstance(images, np.ndarray) and images.ndim == 3:
            images = [images]
        all_encodings = []
        for metric, imgs in zip(self.encoders, tee(images, len(self.encoders))):
            saved = metric.flatten
            metric.flatten = True          # every member is flattened: output sizes differ between encoders
            try:
                all_encodings.append(metric.encode(imgs))
            finally:
                metric.flatten = saved
        return np.hstack(all_encodings)

    def generate_encoding_map(self, image_paths: Iterable[str]) -> dict[str, np.ndarray]:
        import cv2
        image_paths = tuple(image_paths)
        images = (cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB) for path in image_paths)
        return dict(zip(image_paths, self.encode(images)))

    @property
    def similarity_func(self):
        return self._similarity_func

    @similarity_func.setter
    def similarity_func(self, func):
        if func is cosine_similarity:
            self._similarity_func = func
            return
        dummy1, dummy2 = np.random.rand(10, 10), np.random.rand(10, 10)
        self._similarity_func = check_desired_output(func, dummy1, dummy2)

    def similarity_score(self, images1, images2) -> np.ndarray:
        return np.float32(self.similarity_func(self.encode(images1), self.encode(images2)))

    def __repr__(self) -> str:
        encoders_str = "\n".join(str(encoder) for encoder in self.encoders)
        name = getattr(self._similarity_func, "__name__", str(self._similarity_func))
        return f"Pipeline(\nencoders=[{encoders_str}],\nsimilarity_func={name})"
