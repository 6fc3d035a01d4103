"""Plain-array stand-ins for the scikit-learn objects the reference plugs into its encoders, so that a
codebook / GMM / PCA can be supplied without scikit-learn (it may be absent where the engine runs).
They carry exactly the attributes the encoders read ('Plug-in duck types')."""
from __future__ import annotations

import numpy as np

__all__ = ["KMeansModel", "GMMModel", "PCAModel", "save_model", "load_model"]


class KMeansModel:
    def __init__(self, cluster_centers):
        self.cluster_centers_ = np.ascontiguousarray(cluster_centers, dtype=np.float32)
        if self.cluster_centers_.ndim != 2:
            raise ValueError("cluster_centers must be (K, D)")
        self.n_clusters, self.n_features_in_ = self.cluster_centers_.shape


class GMMModel:
    covariance_type = "diag"

    def __init__(self, weights, means, covariances):
        self.weights_ = np.ascontiguousarray(weights, dtype=np.float64)
        self.means_ = np.ascontiguousarray(means, dtype=np.float64)
        self.covariances_ = np.ascontiguousarray(covariances, dtype=np.float64)
        if self.means_.ndim != 2 or self.covariances_.shape != self.means_.shape:
            raise ValueError("means and diagonal covariances must both be (K, D)")
        self.n_components, self.n_features_in_ = self.means_.shape


class PCAModel:
    whiten = False

    def __init__(self, components, mean):
        self.components_ = np.ascontiguousarray(components, dtype=np.float32)
        self.mean_ = np.ascontiguousarray(mean, dtype=np.float32).reshape(-1)
        self.n_components, self.n_features_in_ = self.components_.shape


def save_model(path: str, model) -> None:
    """Persist a fitted clustering / PCA model (scikit-learn object or one of the classes above) as a plain
    .npz of arrays -- the engine's replacement for the reference's joblib pickles (nothing executable)."""
    if hasattr(model, "cluster_centers_"):
        np.savez(path, kind="kmeans", cluster_centers=np.asarray(model.cluster_centers_, np.float32))
    elif hasattr(model, "means_"):
        np.savez(path, kind="gmm", weights=model.weights_, means=model.means_, covariances=model.covariances_)
    elif hasattr(model, "components_"):
        np.savez(path, kind="pca", components=np.asarray(model.components_, np.float32),
                 mean=np.asarray(model.mean_, np.float32))
    else:
        raise ValueError(f"cannot serialise {type(model)}")


def load_model(path: str):
    with np.load(path, allow_pickle=False) as z:
        kind = str(z["kind"])
        if kind == "kmeans":
            m = KMeansModel(z["cluster_centers"])
            if "derived_from" in z.files:          # a stand-in codebook (see tests/golden/extract_reference_tables.py)
                m.derived_from = str(z["derived_from"])
            return m
        if kind == "gmm":
            return GMMModel(z["weights"], z["means"], z["covariances"])
        if kind == "pca":
            return PCAModel(z["components"], z["mean"])
    raise ValueError(f"{path}: unknown model kind {kind!r}")
