"""Seeded inputs of the shipped-tables fixture (tests/golden/shipped_tables.npz), shared by its generator and the tests."""
import os

import numpy as np

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "python-visual-similarity_amd", "pvsim",
                      "res", "model_files")


def shipped_inputs():
    """Seeded inputs shared by this generator and the tests: (raws: list of (n, 128) uint8 SIFT-like images, deep: list of
    (196, 514) float32 deep-feature-like images)."""
    from pvsim import synth
    rng = np.random.default_rng(20261)
    proto = synth.sift_prototypes()
    raws = [synth.sift_like(n, rng, proto).astype(np.uint8) for n in (7, 200, 512, 1257, 33)]
    g = np.load(os.path.join(MODELS, "gmm_k256_deep_features_vgg16_pca.npz"), allow_pickle=False)
    p = np.load(os.path.join(MODELS, "pca_k256_deep_features_vgg16_f2.npz"), allow_pickle=False)
    deep = []
    for _ in range(4):
        comp = rng.choice(256, size=196, p=g["weights"])
        z = g["means"][comp] + np.sqrt(g["covariances"][comp]) * rng.standard_normal((196, 257))
        x = z @ p["components"].astype(np.float64) + p["mean"].astype(np.float64) + 0.05 * rng.standard_normal((196, 514))
        deep.append(x.astype(np.float32))
    return raws, deep


