"""Golden fits for vocabulary training (ImageEncoderBase.learn).

Runs the REFERENCE's own learn() (pyvisim/encoders/_base_encoder.py:311-342) -- imported from /root/reference with the
same stand-ins as make_golden.py, nothing copied -- on seeded synthetic descriptors, with explicit starting points so
that no random draw is involved, and records the tables scikit-learn (pinned in versions.json) fits.  Writes
tests/golden/learn_*.npz.  Run:  python tests/golden/make_golden_learn.py
"""
import io
import json
import os
import sys
import warnings
from contextlib import redirect_stdout

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference  # noqa: E402


def blobs(n, D, K, seed):
    """uint8-valued descriptors (stored compactly, exact in fp32): K overlapping blobs"""
    rng = np.random.default_rng(seed)
    mu = rng.uniform(40, 200, size=(K, D))
    sd = rng.uniform(8, 30, size=(K, 1))
    z = rng.integers(0, K, size=n)
    x = mu[z] + sd[z] * rng.standard_normal((n, D))
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def main():
    import sklearn
    VLADEncoder, FisherVectorEncoder, Pipeline, Lambda, ref_cos, ref_eval = _import_reference()
    warnings.simplefilter("ignore")
    n_img, per, D, K = 40, 500, 32, 16
    xu8 = blobs(n_img * per, D, K, 4242)
    images = [xu8[i * per:(i + 1) * per].astype(np.int64) for i in range(n_img)]   # 2-D integer "mask" images
    fx = Lambda(lambda im: (im.astype(np.float32) / 16.0), D)                        # descriptors = x / 16 (fp32)
    X = xu8.astype(np.float32) / 16.0
    rng = np.random.default_rng(7)
    C0 = X[rng.choice(len(X), K, replace=False)].copy()
    out = {"x_u8": xu8, "c0": C0}

    def learn(enc, **kw):
        with redirect_stdout(io.StringIO()):
            enc.learn(images, **kw)
        return enc

    # ---- k-means: 3 fixed Lloyd iterations, and the default stopping rule
    from sklearn.cluster import KMeans
    dummy = KMeans(n_clusters=K, init=C0, n_init=1, max_iter=1).fit(X)
    e = learn(VLADEncoder(fx, kmeans_model=dummy), n_clusters=K, init=C0.copy(), n_init=1, max_iter=3, tol=0.0)
    km = e.clustering_model
    out.update(km3_centers=km.cluster_centers_, km3_labels=km.labels_.astype(np.int32), km3_inertia=np.float64(km.inertia_),
               km3_n_iter=np.int64(km.n_iter_))
    e = learn(VLADEncoder(fx, kmeans_model=dummy), n_clusters=K, init=C0.copy(), n_init=1)
    km = e.clustering_model
    out.update(km_centers=km.cluster_centers_, km_labels=km.labels_.astype(np.int32), km_inertia=np.float64(km.inertia_),
               km_n_iter=np.int64(km.n_iter_))
    print("  kmeans default: n_iter", km.n_iter_, "inertia", km.inertia_)

    # ---- PCA + k-means through dim_reduction_factor=2
    e = learn(VLADEncoder(fx, kmeans_model=dummy), n_clusters=K, dim_reduction_factor=2, init=None or "k-means++", n_init=1,
              random_state=0, max_iter=1)
    out.update(pca_components=e.pca.components_, pca_mean=e.pca.mean_, pca_explained_variance=e.pca.explained_variance_)

    # ---- GMM: 5 fixed EM iterations and the default stopping rule, explicit start
    from sklearn.mixture import GaussianMixture
    Kg = 8
    w0 = np.full(Kg, 1.0 / Kg)
    m0 = C0[:Kg].astype(np.float64)
    p0 = np.full((Kg, D), 1.0 / 4.0)
    gd = GaussianMixture(Kg, covariance_type="diag", weights_init=w0, means_init=m0, precisions_init=p0, max_iter=1).fit(X)
    out.update(g_w0=w0, g_m0=m0, g_p0=p0)
    e = learn(FisherVectorEncoder(fx, gmm_model=gd), n_clusters=Kg, weights_init=w0, means_init=m0, precisions_init=p0,
              max_iter=5, tol=0.0)
    g = e.clustering_model
    out.update(g5_weights=g.weights_, g5_means=g.means_, g5_cov=g.covariances_, g5_lower=np.float64(g.lower_bound_),
               g5_n_iter=np.int64(g.n_iter_))
    e = learn(FisherVectorEncoder(fx, gmm_model=gd), n_clusters=Kg, weights_init=w0, means_init=m0, precisions_init=p0)
    g = e.clustering_model
    out.update(g_weights=g.weights_, g_means=g.means_, g_cov=g.covariances_, g_lower=np.float64(g.lower_bound_),
               g_n_iter=np.int64(g.n_iter_), g_converged=np.bool_(g.converged_))
    print("  gmm default: n_iter", g.n_iter_, "converged", g.converged_, "lower bound", g.lower_bound_)

    path = os.path.join(HERE, "learn_k16_d32.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote learn_k16_d32.npz ({os.path.getsize(path) / 1e6:.2f} MB)")
    vj = os.path.join(HERE, "versions.json")
    meta = json.load(open(vj))
    meta["files"]["learn_k16_d32"] = {k: [list(np.shape(v)), str(np.asarray(v).dtype)] for k, v in out.items()}
    meta["learn_generated_with"] = {"scikit-learn": sklearn.__version__, "numpy": np.__version__}
    json.dump(meta, open(vj, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
