"""FisherVectorEncoder -- drop-in for pyvisim/encoders/fisher_vector.py:15-135, computed by csrc/fisher.hip.

Per image: descriptors -> (PCA) -> diagonal-GMM posteriors -> 0th/1st/2nd-order moments -> gradients wrt
(pi, mu, sigma^2) with the analytic Fisher normalisation -> sign|v|^p (p = 0.5) -> global L_ord normalisation.
Output (N, K + 2KD) float64, laid out [d_pi | d_mu (k-major) | d_sigma], like the reference."""
from __future__ import annotations

import warnings
from typing import Callable

import numpy as np

from ..engine import DESC_F32

from ..features._features import FeatureExtractorBase, RootSIFT
from .._utils import cosine_similarity
from ._base_encoder import ImageEncoderBase


def _is_gmm_like(model) -> bool:
    if type(model).__name__ == "GaussianMixture" and type(model).__module__.startswith("sklearn."):
        return True
    return all(hasattr(model, a) for a in ("weights_", "means_", "covariances_", "n_features_in_"))


class FisherVectorEncoder(ImageEncoderBase):
    """:param gmm_model: fitted sklearn GaussianMixture (diag), or pvsim.models.GMMModel (plain arrays)
    other parameters as in the reference (fisher_vector.py:41-51)"""

    def __init__(self, feature_extractor: FeatureExtractorBase = None, weights=None, gmm_model=None,
                 power_norm_weight: float = 0.5, norm_order: int = 2, epsilon: float = 1e-9, flatten: bool = True,
                 similarity_func: Callable[[np.ndarray, np.ndarray], float] = cosine_similarity, pca=None,
                 raise_error_when_pca_incompatible: bool = False, **engine_kwargs):
        if feature_extractor is None:
            feature_extractor = RootSIFT()
        if gmm_model is not None:
            if not _is_gmm_like(gmm_model):
                raise ValueError(f"The clustering model must be an instance of GaussianMixture, not {type(gmm_model)}")
            gmm_model.covariance_type = "diag"
        if weights is not None and weights.__class__.__name__ != "GMMWeights":
            raise ValueError(f"You can only pass an instance of GMMWeights, not {weights.__class__.__name__}")
        super().__init__(feature_extractor, weights, gmm_model, similarity_func, power_norm_weight, norm_order,
                         epsilon, flatten, pca, raise_error_when_pca_incompatible, **engine_kwargs)

    @property
    def clustering_model(self):
        return ImageEncoderBase.clustering_model.fget(self)

    @clustering_model.setter
    def clustering_model(self, model):
        if not _is_gmm_like(model):
            raise ValueError(f"The clustering model must be an instance of GaussianMixture, not {type(model)}")
        if getattr(model, "covariance_type", "diag") != "diag":
            warnings.warn("Attribute 'covariance_type' of the clustering model is set to 'diag' because training "
                          "will take too long otherwise.")
            model.covariance_type = "diag"
        if np.asarray(model.covariances_).shape != np.asarray(model.means_).shape:
            raise ValueError("only diagonal covariances (K, D) are supported")
        ImageEncoderBase.clustering_model.fset(self, model)

    def _make_tables(self):
        m = self._clustering_model
        return self.context.gmm(np.asarray(m.weights_), np.asarray(m.means_), np.asarray(m.covariances_)), self._pca_table()

    def _encode_packed(self, packed, offsets, kind):
        g, pca = self._device_tables()
        return self.context.fisher_encode(g, packed, offsets, kind, self.power_norm_weight, self.norm_order,
                                          self.epsilon, pca)

    def _encode_device(self, d_desc, d_offsets, n_images, total_desc):
        g, pca = self._device_tables()
        ctx = self.context
        L = g.K + 2 * g.K * g.D
        buf = ctx.buffer(n_images * L * 8)
        try:
            ctx.fisher_encode_dev(g, d_desc, DESC_F32, d_offsets, n_images, total_desc, buf.ptr, 1, self.power_norm_weight,
                                  self.norm_order, self.epsilon, pca)
            return buf.download((n_images, L), np.float64)      # blocks until the result is on the host
        finally:
            buf.free()

    def _empty_quirk(self):
        raise ZeroDivisionError("an image without descriptors has no Fisher vector (reference divides by zero, "
                                "fisher_vector.py:93-104)")
