"""Seeded inputs of the BASELINE configs[2] shape, shared by make_golden.py (which feeds them to the reference) and the tests."""
import numpy as np


def config3_inputs(n_images=12):
    """BASELINE configs[2] shape (Fisher, D = 512, K = 256, n = 196): seeded synthetic GMM tables and descriptors, the same
    on every machine (numpy's Generator streams are version-stable); the fixture fisher_k256_d512.npz stores the reference's outputs for them."""
    rng = np.random.default_rng(1236)
    K, D, n = 256, 512, 196
    means = rng.normal(0.0, 2.0, size=(K, D))
    cov = np.exp(rng.uniform(np.log(1e-3), np.log(25.0), size=(K, D)))
    w = rng.dirichlet(np.full(K, 5.0))
    imgs = []
    for _ in range(n_images):
        z = rng.integers(0, K, size=n)
        imgs.append((means[z] + np.sqrt(cov[z]) * rng.standard_normal((n, D))).astype(np.float32))
    return w, means, cov, imgs
