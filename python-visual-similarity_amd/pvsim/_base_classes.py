"""The two abstract interfaces of the reference (pyvisim/_base_classes.py:9-54)."""
from __future__ import annotations

import abc
import logging
from typing import Iterable

import numpy as np

from ._utils import is_numpy_image


class SimilarityMetric(abc.ABC):
    """Base of everything that can score two (batches of) images."""
    _logger = logging.getLogger("Similarity_Metrics")

    @abc.abstractmethod
    def similarity_score(self, image1: Iterable[np.ndarray], image2: Iterable[np.ndarray]):
        ...


class FeatureExtractorBase(abc.ABC):
    """image (NumPy array) -> local descriptors (n, output_dim)."""
    _logger = logging.getLogger("Feature_Extractor")

    def __init__(self):
        pass

    @abc.abstractmethod
    def __call__(self, image: np.ndarray):
        # every extractor validates its input first (reference: _base_classes.py:46)
        is_numpy_image(image, 0)

    @property
    @abc.abstractmethod
    def output_dim(self) -> int:
        ...
