"""pvsim -- an MI355X-native image-similarity engine with pyvisim's encoder API.

    from pvsim.encoders import VLADEncoder, FisherVectorEncoder, Pipeline, KMeansWeights, GMMWeights
    from pvsim.features import RootSIFT, SIFT, Lambda, DeepConvFeature
    from pvsim import eval                     # retrieve_top_k_similar, top_k_map, top_k_accuracy

The arithmetic (centroid assignment, VLAD / Fisher aggregation, normalisation, cosine GEMM, top-k) runs in
hand-written HIP kernels for gfx950 behind a C-ABI (include/pvsim.h) bound with ctypes; there is no CPU
fallback.  Importing the package does not touch the GPU; the first computation does.
"""
from .engine import Context, default_context, pack_descriptors
from . import models

__version__ = "0.1.0"
__all__ = ["encoders", "features", "eval", "models", "Context", "default_context", "pack_descriptors"]
