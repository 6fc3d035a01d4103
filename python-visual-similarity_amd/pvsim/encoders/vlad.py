"""VLADEncoder -- drop-in for pyvisim/encoders/vlad.py:12-115, computed by the HIP kernels of csrc/vlad.hip.

Per image: descriptors -> (PCA) -> nearest centroid (KMeans.predict semantics) -> per-cluster residual sums in
descriptor order -> sign|v|^p -> per-cluster L_ord normalisation (+eps) -> k-major flatten.  fp32 throughout."""
from __future__ import annotations

from typing import Callable

import numpy as np

from ..engine import DESC_F32

from ..features._features import FeatureExtractorBase, RootSIFT
from .._utils import cosine_similarity
from ._base_encoder import ImageEncoderBase


def _is_kmeans_like(model) -> bool:
    if type(model).__name__ == "KMeans" and type(model).__module__.startswith("sklearn."):
        return True
    return hasattr(model, "cluster_centers_") and hasattr(model, "n_features_in_")


class VLADEncoder(ImageEncoderBase):
    """:param feature_extractor: FeatureExtractorBase instance (default RootSIFT)
    :param weights: a KMeansWeights member (pretrained codebook) -- overrides kmeans_model / pca
    :param kmeans_model: fitted sklearn KMeans, or pvsim.models.KMeansModel (plain centroid array)
    :param power_norm_weight, norm_order, epsilon, flatten, similarity_func, pca,
           raise_error_when_pca_incompatible: as in the reference (vlad.py:42-53)"""

    def __init__(self, feature_extractor: FeatureExtractorBase = None, weights=None, kmeans_model=None,
                 power_norm_weight: float = 1, norm_order: int = 2, epsilon: float = 1e-9, flatten: bool = True,
                 similarity_func: Callable[[np.ndarray, np.ndarray], float] = cosine_similarity, pca=None,
                 raise_error_when_pca_incompatible: bool = False, **engine_kwargs):
        if feature_extractor is None:
            feature_extractor = RootSIFT()
        if kmeans_model is not None and not _is_kmeans_like(kmeans_model):
            raise ValueError(f"The clustering model must be an instance of KMeans, not {type(kmeans_model)}")
        if weights is not None and weights.__class__.__name__ != "KMeansWeights":
            raise ValueError(f"You can only pass an instance of KMeansWeights, not {weights.__class__.__name__}")
        super().__init__(feature_extractor, weights, kmeans_model, similarity_func, power_norm_weight, norm_order,
                         epsilon, flatten, pca, raise_error_when_pca_incompatible, **engine_kwargs)

    @property
    def clustering_model(self):
        return ImageEncoderBase.clustering_model.fget(self)

    @clustering_model.setter
    def clustering_model(self, model):
        if not _is_kmeans_like(model):
            raise ValueError(f"The clustering model must be an instance of KMeans, not {type(model)}")
        ImageEncoderBase.clustering_model.fset(self, model)

    def _make_tables(self):
        return self.context.codebook(np.asarray(self._clustering_model.cluster_centers_)), self._pca_table()

    def _encode_packed(self, packed, offsets, kind):
        cb, pca = self._device_tables()
        return self.context.vlad_encode(cb, packed, offsets, kind, self.power_norm_weight, self.norm_order,
                                        self.epsilon, pca)

    def _encode_device(self, d_desc, d_offsets, n_images, total_desc):
        cb, pca = self._device_tables()
        ctx = self.context
        L = cb.K * cb.D
        buf = ctx.buffer(n_images * L * 4)
        try:
            ctx.vlad_encode_dev(cb, d_desc, DESC_F32, d_offsets, n_images, total_desc, buf.ptr, self.power_norm_weight,
                                self.norm_order, self.epsilon, pca)
            return buf.download((n_images, L), np.float32)
        finally:
            buf.free()

    def _shape_output(self, out):
        if self.flatten:
            return out
        k, d = self._clustering_model.cluster_centers_.shape
        return out.reshape(out.shape[0] * k, d)        # np.vstack of (K, D) blocks (vlad.py:110-115)

    def _empty_quirk(self):
        k, d = self._clustering_model.cluster_centers_.shape
        return np.zeros(k * d, dtype=np.float32)        # vlad.py:92-93

    def predict(self, descriptors) -> np.ndarray:
        """KMeans.predict labels (int32) of a packed (n, D) descriptor array, on the device."""
        cb, pca = self._device_tables()
        x = np.ascontiguousarray(descriptors, dtype=np.float32)
        _, labels = self.context.vlad_encode(cb, x, np.array([0, x.shape[0]], np.int64), 0, 1.0, 2, 1e-9, pca,
                                             return_labels=True)
        return labels
