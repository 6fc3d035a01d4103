"""Encoder base: constructor / setter validation, `similarity_score`, `generate_encoding_map`, `learn` -- the
drop-in boundary of pyvisim/encoders/_base_encoder.py:158-401 -- plus the glue that turns the plugged-in
clustering / PCA objects into device tables and sends packed descriptor batches through the C-ABI."""
from __future__ import annotations

import abc
import os
import warnings
from collections.abc import Iterator, MutableSequence
from enum import Enum
from functools import wraps
from typing import Any, Callable, Iterable, Optional

import numpy as np

from .._base_classes import FeatureExtractorBase, SimilarityMetric
from .. import models as _models
from ..engine import DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT, default_context, pack_descriptors

MODEL_FILES_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "res", "model_files")


# ----------------------------------------------------------------------------------------- helpers
def check_desired_output(similarity_func: Callable[[np.ndarray, np.ndarray], Any], vecs1: np.ndarray,
                         vecs2: np.ndarray) -> Callable[[np.ndarray, np.ndarray], np.ndarray]:
    """Runs a user similarity function once; if it raises, does not return an ndarray, or does not return
    (len(vecs1), len(vecs2)), it is wrapped in a row-by-row loop (reference: _base_encoder.py:23-97)."""
    try:
        out = similarity_func(vecs1, vecs2)
    except Exception as e:
        warnings.warn(f"Similarity function threw an error: {e}. Falling back to row-wise loop.")
        return _make_fallback_func(similarity_func)
    if not isinstance(out, np.ndarray):
        warnings.warn(f"Expected a NumPy array, got {type(out)}. Using fallback method.")
        return _make_fallback_func(similarity_func)
    ok = True
    if out.ndim == 2:
        ok = out.shape[0] == vecs1.shape[0] and out.shape[1] == vecs2.shape[0]
    elif out.ndim == 1 and out.size != 1:
        ok = False
    if not ok:
        warnings.warn(f"Output shape {out.shape} is not the expected (N, M). Expected output shape to be "
                      f"({vecs1.shape[0]}, {vecs2.shape[0]}). Using fallback.")
        return _make_fallback_func(similarity_func)
    return similarity_func


def _make_fallback_func(sim_func):
    def fallback(vecs1: np.ndarray, vecs2: np.ndarray) -> np.ndarray:
        out = np.zeros((vecs1.shape[0], vecs2.shape[0]), dtype=np.float32)
        for i in range(vecs1.shape[0]):
            for j in range(vecs2.shape[0]):
                out[i, j] = sim_func(vecs1[i:i + 1], vecs2[j:j + 1])
        return out

    fallback.__name__ = getattr(sim_func, "__name__", "fallback")
    return fallback


def _tupleize_first_arg(func: Callable) -> Callable:
    @wraps(func)
    def wrapper(self, image_paths: Any, /, *args, **kwargs):
        if isinstance(image_paths, (Iterator, MutableSequence)):
            image_paths = tuple(image_paths)
        return func(self, image_paths, *args, **kwargs)

    return wrapper


# ----------------------------------------------------------------------------------------- pretrained tables
class _PretrainedModels(Enum):
    """Same member names as the reference enums (pyvisim/encoders/_base_encoder.py:117-155).  The reference stores joblib
    pickles of scikit-learn objects; this engine stores plain arrays (`.npz`, pvsim.models.save_model) and never unpickles
    anything.  The five GaussianMixture and three PCA tables the reference ships are here as arrays, read out of its files by
    tests/golden/extract_reference_tables.py (an opcode walk that executes nothing).  The reference's KMeans files -- and its
    VGG16 no-PCA mixture -- are absent from its own checkout: the KMeansWeights members load the float32 means_ of the
    matching mixture as a stand-in codebook (with a warning), GMMWeights.OXFORD102_K256_VGG16 raises FileNotFoundError."""

    def load(self) -> object:
        if not os.path.exists(self.value):
            raise FileNotFoundError(
                f"{self.value} not found (the reference's checkout does not hold this model either). Convert a fitted model "
                f"with pvsim.models.save_model(path, model): plain arrays, nothing is unpickled.")
        m = _models.load_model(self.value)
        if getattr(m, "derived_from", None):
            warnings.warn(f"{self.name}: the reference's own KMeans file is absent from its checkout; this codebook is the "
                          f"{m.derived_from}", stacklevel=2)
        return m


class KMeansWeights(_PretrainedModels):
    OXFORD102_K256_VGG16_PCA = f"{MODEL_FILES_PATH}/k_means_k256_deep_features_vgg16_pca.npz"
    OXFORD102_K256_VGG16 = f"{MODEL_FILES_PATH}/k_means_k256_deep_features_vgg16_no_pca.npz"
    OXFORD102_K256_ROOTSIFT_PCA = f"{MODEL_FILES_PATH}/k_means_k256_root_sift_pca.npz"
    OXFORD102_K256_ROOTSIFT = f"{MODEL_FILES_PATH}/k_means_k256_root_sift_no_pca.npz"
    OXFORD102_K256_SIFT_PCA = f"{MODEL_FILES_PATH}/k_means_k256_sift_pca.npz"
    OXFORD102_K256_SIFT = f"{MODEL_FILES_PATH}/k_means_k256_sift_no_pca.npz"


class _PCA(_PretrainedModels):
    OXFORD102_PCA256_VGG16 = f"{MODEL_FILES_PATH}/pca_k256_deep_features_vgg16_f2.npz"
    OXFORD102_PCA256_ROOTSIFT = f"{MODEL_FILES_PATH}/pca_k256_root_sift_f2.npz"
    OXFORD102_PCA256_SIFT = f"{MODEL_FILES_PATH}/pca_k256_sift_f2.npz"


class GMMWeights(_PretrainedModels):
    OXFORD102_K256_VGG16_PCA = f"{MODEL_FILES_PATH}/gmm_k256_deep_features_vgg16_pca.npz"
    OXFORD102_K256_VGG16 = f"{MODEL_FILES_PATH}/gmm_k256_deep_features_vgg16_no_pca.npz"
    OXFORD102_K256_ROOTSIFT_PCA = f"{MODEL_FILES_PATH}/gmm_k256_root_sift_pca.npz"
    OXFORD102_K256_ROOTSIFT = f"{MODEL_FILES_PATH}/gmm_k256_root_sift_no_pca.npz"
    OXFORD102_K256_SIFT_PCA = f"{MODEL_FILES_PATH}/gmm_k256_sift_pca.npz"
    OXFORD102_K256_SIFT = f"{MODEL_FILES_PATH}/gmm_k256_sift_no_pca.npz"


_CLUSTERING_TO_PCA_MAPPING = {
    KMeansWeights.OXFORD102_K256_VGG16_PCA: _PCA.OXFORD102_PCA256_VGG16,
    KMeansWeights.OXFORD102_K256_ROOTSIFT_PCA: _PCA.OXFORD102_PCA256_ROOTSIFT,
    KMeansWeights.OXFORD102_K256_SIFT_PCA: _PCA.OXFORD102_PCA256_SIFT,
    GMMWeights.OXFORD102_K256_VGG16_PCA: _PCA.OXFORD102_PCA256_VGG16,
    GMMWeights.OXFORD102_K256_ROOTSIFT_PCA: _PCA.OXFORD102_PCA256_ROOTSIFT,
    GMMWeights.OXFORD102_K256_SIFT_PCA: _PCA.OXFORD102_PCA256_SIFT,
}


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and type(x).__name__ == "Tensor"


# ----------------------------------------------------------------------------------------- base class
class ImageEncoderBase(SimilarityMetric):
    """feature extractor -> (PCA) -> clustering model -> aggregated, normalised vector, on the MI355X.

    Constructor arguments, defaults and validation follow the reference class; `strict_compat=True`
    additionally reproduces its empty-image quirk (a batch containing an image without descriptors returns
    ONE 1-D zero vector, vlad.py:92-93) instead of a zero row for that image."""

    def __init__(self, feature_extractor: FeatureExtractorBase = None, weights=None, clustering_model=None,
                 similarity_func: Callable[[np.ndarray, np.ndarray], float] = None, power_norm_weight: float = 1,
                 norm_order: int = 2, epsilon: float = 1e-9, flatten: bool = True, pca=None,
                 raise_error_when_pca_incompatible: bool = True, context=None, strict_compat: bool = False):
        self._feature_extractor = None
        self._clustering_model = None
        self._pca = None
        self._similarity_func = None
        self._ctx = context
        self._tables = None            # (device clustering table, device pca table) cache
        # (the reference reads this flag in the clustering setter before assigning it -- A.3; fixed here)
        self.raise_error_when_pca_incompatible = raise_error_when_pca_incompatible
        self.strict_compat = strict_compat

        self.similarity_func = similarity_func
        self.feature_extractor = feature_extractor
        if weights is not None:
            if "PCA" in weights.name:
                self.pca = _CLUSTERING_TO_PCA_MAPPING[weights].load()
            self.clustering_model = weights.load()
        else:
            if pca is not None:
                self.pca = pca
            if clustering_model is not None:
                self.clustering_model = clustering_model
        self.power_norm_weight = power_norm_weight
        self.norm_order = norm_order
        self.epsilon = epsilon
        self.flatten = flatten

    # ---- plug-in properties (validation as in _base_encoder.py:222-309)
    @property
    def context(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    @property
    def feature_extractor(self) -> FeatureExtractorBase:
        return self._feature_extractor

    @feature_extractor.setter
    def feature_extractor(self, feature_extractor: FeatureExtractorBase):
        if not isinstance(feature_extractor, FeatureExtractorBase):
            raise TypeError(f"feature_extractor must be an instance of FeatureExtractorBase, not {type(feature_extractor)}")
        if self._pca is not None:
            if feature_extractor.output_dim != self._pca.n_features_in_:
                raise RuntimeError(f"Feature Extractor outputs shape {feature_extractor.output_dim}, "
                                   f"But PCA accepts input dim {self._pca.n_features_in_}")
        elif self._clustering_model is not None:
            if feature_extractor.output_dim != self._clustering_model.n_features_in_:
                raise RuntimeError(f"Feature Extractor outputs shape {feature_extractor.output_dim}, "
                                   f"But clustering model accepts input dim {self._clustering_model.n_features_in_}")
        self._feature_extractor = feature_extractor

    @property
    def similarity_func(self):
        return self._similarity_func

    @similarity_func.setter
    def similarity_func(self, func):
        if func is None:
            raise TypeError("similarity_func must be callable")
        from .._utils import cosine_similarity
        if func is cosine_similarity:
            # the built-in needs a GPU; its (N, M) contract is known, so it is not probed at construction
            self._similarity_func = func
            return
        dummy1, dummy2 = np.random.rand(10, 10), np.random.rand(10, 10)
        self._similarity_func = check_desired_output(func, dummy1, dummy2)

    @property
    def clustering_model(self):
        return self._clustering_model

    @clustering_model.setter
    def clustering_model(self, clustering_model):
        if self._pca is not None:
            if self._pca.n_components != clustering_model.n_features_in_:
                msg = (f"PCA is incompatible with the new clustering model. PCA input size: {self._pca.n_components}, "
                       f"New clustering model input size: {clustering_model.n_features_in_}. ")
                if self.raise_error_when_pca_incompatible:
                    raise RuntimeError(msg + "If you want the PCA to be reset to None instead, set "
                                             "raise_error_when_pca_incompatible=False.")
                warnings.warn(msg + "PCA will be reset to None to avoid errors.")
                self._pca = None
        elif self._feature_extractor.output_dim != clustering_model.n_features_in_:
            raise RuntimeError("Feature extractor output size has to match the clustering model input size. "
                               f"Feature extractor has output size {self._feature_extractor.output_dim}, "
                               f"while clustering model has input size {clustering_model.n_features_in_}")
        self._clustering_model = clustering_model
        self._tables = None

    @property
    def pca(self):
        return self._pca

    @pca.setter
    def pca(self, pca):
        if pca.n_features_in_ != self._feature_extractor.output_dim:
            raise ValueError("PCA input size has to match the feature extractor output size. "
                             f"PCA model has input size {pca.n_features_in_}, "
                             f"while feature extractor has output size {self._feature_extractor.output_dim}")
        if self._clustering_model is not None and pca.n_components != self._clustering_model.n_features_in_:
            raise ValueError("PCA input size has to match the clustering model input size."
                             f"PCA model has input size {pca.n_components}, "
                             f"while clustering model has input size {self._clustering_model.n_features_in_}")
        self._pca = pca
        self._tables = None

    # ---- device glue
    def _pca_table(self):
        if self._pca is None:
            return None
        if getattr(self._pca, "whiten", False):
            raise NotImplementedError("whitened PCA is not supported (the reference's models use whiten=False)")
        return self.context.pca(np.asarray(self._pca.components_), np.asarray(self._pca.mean_))

    @abc.abstractmethod
    def _make_tables(self):
        """-> (clustering table handle, pca table handle or None)"""

    def _device_tables(self):
        if self._clustering_model is None:
            raise RuntimeError("no clustering model set: pass one to the constructor or call learn()")
        if self._tables is None:
            self._tables = self._make_tables()
        return self._tables

    @abc.abstractmethod
    def _encode_packed(self, packed: np.ndarray, offsets: np.ndarray, kind: int) -> np.ndarray:
        """(sum n, D_in) descriptors + CSR offsets -> (N, L) encodings, through the C-ABI."""

    @property
    def _input_dim(self) -> int:
        return self._pca.n_features_in_ if self._pca is not None else self._clustering_model.n_features_in_

    def encode_descriptors(self, descriptors, offsets=None, *, rootsift: bool = False) -> np.ndarray:
        """Descriptor-level entry point (the timed configs start here).

        `descriptors`: a list of (n_i, D) arrays, or one packed (sum n_i, D) array together with
        `offsets` (int64, N+1).  rootsift=True means the rows are RAW SIFT (uint8 or float32, 0..255) and
        the RootSIFT transform is fused into the GPU load (uint8 rows cost 4x fewer HBM bytes)."""
        d_in = self._input_dim
        if offsets is None:
            lst = list(descriptors)
            u8 = rootsift and all(d.dtype == np.uint8 for d in lst if d is not None)
            packed, offsets = pack_descriptors(lst, d_in, np.uint8 if u8 else np.float32)
        else:
            packed = np.asarray(descriptors)
            u8 = rootsift and packed.dtype == np.uint8
        kind = (DESC_U8_ROOTSIFT if u8 else DESC_F32_ROOTSIFT) if rootsift else DESC_F32
        out = self._encode_packed(packed, np.asarray(offsets, dtype=np.int64), kind)
        return self._shape_output(out)

    def _shape_output(self, out: np.ndarray) -> np.ndarray:
        return out

    def _gather_descriptors(self, images):
        """Run the extractor per image (host side, as the reference does) -> (list of descriptors, kind)."""
        if _is_torch_tensor(images):
            raise RuntimeError("Torch images are not supported yet.")
        if isinstance(images, np.ndarray) and images.ndim == 3:
            images = [images]
        fx = self.feature_extractor
        fused = getattr(fx, "fused_rootsift", False) and self._pca is None and hasattr(fx, "raw")
        descs = []
        for image in images:
            descs.append(fx.raw(image) if fused else fx(image))
        return descs, (DESC_F32_ROOTSIFT if fused else DESC_F32)

    def _device_features(self, images):
        """Extractors that keep their features on the GPU (`batch(images)` -> (B, n, D) float32 CUDA tensor, e.g.
        DeepConvFeature) hand them to the encoder kernels as a device pointer: no per-image `.cpu().numpy()` round trip
        (the reference copies every image's feature map to the host, features/_features.py:276-289)."""
        fx = self.feature_extractor
        dev = getattr(fx, "device", None)
        if not hasattr(fx, "batch") or dev is None or getattr(dev, "type", "cpu") != "cuda":
            return None
        if _is_torch_tensor(images):
            raise RuntimeError("Torch images are not supported yet.")
        if isinstance(images, np.ndarray) and images.ndim == 3:
            images = [images]
        images = list(images)
        if not images:
            raise ValueError("need at least one array to concatenate")
        import torch
        if (dev.index or 0) != self.context.device:
            return None
        outs = []
        for c0 in range(0, len(images), 64):
            feats = fx.batch(images[c0:c0 + 64])
            b, n, d = feats.shape
            if d != self._input_dim:
                raise RuntimeError(f"descriptor dimension {d} does not match the model input dimension {self._input_dim}")
            if n == 0:
                return None
            offsets = (torch.arange(b + 1, dtype=torch.int64, device=feats.device) * n).contiguous()
            torch.cuda.current_stream(feats.device).synchronize()      # the extractor ran on torch's stream
            outs.append(self._encode_device(feats.data_ptr(), offsets.data_ptr(), b, b * n))
        return np.vstack(outs)

    def _encode_device(self, d_desc: int, d_offsets: int, n_images: int, total_desc: int) -> np.ndarray:
        raise NotImplementedError

    def encode(self, images: Iterable[np.ndarray] | np.ndarray) -> np.ndarray:
        """(N, L) encodings of one image (H, W, 3) or an iterable of images."""
        if hasattr(self.feature_extractor, "batch"):
            images = [images] if isinstance(images, np.ndarray) and images.ndim == 3 else list(images)
            out = self._device_features(images)
            if out is not None:
                return self._shape_output(out)
        descs, kind = self._gather_descriptors(images)
        if not descs:
            raise ValueError("need at least one array to concatenate")   # np.vstack([]) in the reference
        if self.strict_compat and any(d is None or d.shape[0] == 0 for d in descs):
            return self._empty_quirk()
        packed, offsets = pack_descriptors(descs, self._input_dim, np.float32)
        return self._shape_output(self._encode_packed(packed, offsets, kind))

    def _empty_quirk(self):
        raise NotImplementedError

    # ---- reference API
    def learn(self, images: Iterable[np.ndarray], /, *, n_clusters: int, dim_reduction_factor: int = None,
              **kwargs) -> None:
        """Learns the visual vocabulary (reference: _base_encoder.py:311-342: PCA.fit -> KMeans.fit for VLAD /
        GaussianMixture(covariance_type="diag").fit for Fisher, on the stacked descriptors of `images`).

        The fits run on the device (pvsim/learn.py); `kwargs` are the scikit-learn estimator's keyword arguments
        (KMeans: init, n_init, max_iter, tol, random_state, verbose; GaussianMixture: tol, reg_covar, max_iter, n_init,
        init_params, weights_init, means_init, precisions_init, random_state, verbose).  Anything else raises
        TypeError, as an unknown keyword does in scikit-learn."""
        features = np.vstack([self.feature_extractor(image) for image in images])
        self.learn_from_descriptors(features, n_clusters=n_clusters, dim_reduction_factor=dim_reduction_factor, **kwargs)

    def learn_from_descriptors(self, features, /, *, n_clusters: int, dim_reduction_factor: int = None,
                               rootsift: bool = False, **kwargs) -> None:
        """learn() from an already stacked (n, D) descriptor matrix (rootsift=True: raw SIFT rows, uint8 or float32,
        transformed on the device as in encode_descriptors)."""
        from .. import learn as _learn
        features = np.asarray(features)
        print("[INFO] Learning the visual vocabulary with the following parameters:")
        print("   - Number of clusters:", n_clusters)
        print("   - Feature Extractor used:", self.feature_extractor.__class__.__name__)
        print("   - Dimension of the feature space:", feat_dim := features.shape[1])
        kind = DESC_F32
        if rootsift:
            kind = DESC_U8_ROOTSIFT if features.dtype == np.uint8 else DESC_F32_ROOTSIFT
        if self.__class__.__name__ not in ("VLADEncoder", "FisherVectorEncoder"):
            raise ValueError("Unknown encoder class.")
        rows = _learn.DeviceRows.from_host(self.context, features, kind)
        reduced = None
        try:
            pca = None
            if dim_reduction_factor:
                print("   - New dimension after PCA reduction:", new_dim := feat_dim // dim_reduction_factor)
                pca = _learn.fit_pca(rows, new_dim)
                reduced = rows.transformed(pca)
            data = reduced if reduced is not None else rows
            if self.__class__.__name__ == "VLADEncoder":
                model = _learn.fit_kmeans(data, n_clusters, **kwargs)
            else:
                model = _learn.fit_gmm(data, n_clusters, **kwargs)
        finally:
            rows.free()
            if reduced is not None:
                reduced.free()
        # install the new tables: PCA first when the new model expects the reduced dimension
        self._clustering_model = None
        self._tables = None
        if pca is not None:
            self._pca = pca
        self.clustering_model = model                          # the setter resets an incompatible older PCA, with a warning

    @_tupleize_first_arg
    def generate_encoding_map(self, image_paths: Iterable[str], /) -> dict[str, np.ndarray]:
        """{image_path: encoded_vector} in input order (duplicates collapse, as in a dict)."""
        import cv2  # image decoding is OpenCV's job in the reference too (_base_encoder.py:358)
        images = (cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB) for path in image_paths)
        return dict(zip(image_paths, self.encode(images)))

    def similarity_score(self, images1, images2) -> np.ndarray:
        """np.float32 similarity matrix (N, M) between two (batches of) images."""
        vector1 = self.encode(images1)
        vector2 = self.encode(images2)
        return np.float32(self.similarity_func(vector1, vector2))

    def __repr__(self) -> str:
        n_clusters = None
        if self._clustering_model is not None:
            n_clusters = getattr(self._clustering_model, "n_clusters", None) or getattr(
                self._clustering_model, "n_components", None)
        return (self.__class__.__name__ + f"(feature_extractor={self.feature_extractor.__class__.__name__}, \n"
                f"similarity_func={getattr(self.similarity_func, '__name__', self.similarity_func)}, \n"
                f"Number of cluster={n_clusters}, \nPower Norm Weight={self.power_norm_weight}, \n"
                f"Norm Order={self.norm_order})")
