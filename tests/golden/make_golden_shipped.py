#!/usr/bin/env python3
"""Golden fixture on the reference's OWN shipped vocabularies: the REFERENCE's encoders run on the tables that
tests/golden/extract_reference_tables.py read out of pyvisim/res/model_files/*.pkl (without unpickling them).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_shipped.py [--check]

Build container only (imports /root/reference through make_golden.py's recipe).  The scikit-learn objects are rebuilt around
the arrays -- with the file's own precisions_cholesky_, which is what a pyvisim user's joblib.load would hand to predict_proba
(pyvisim/encoders/_base_encoder.py:117-121) -- and given to the reference's FisherVectorEncoder / VLADEncoder:

  * fisher_rootsift     GMMWeights.OXFORD102_K256_ROOTSIFT: D = 128, 487 covariance entries at the reg_covar floor (precision
                        1e6): the numerically hardest real case of the path (large cancelling terms in the log-density)
  * fisher_rootsift_pca GMMWeights.OXFORD102_K256_ROOTSIFT_PCA + _PCA.OXFORD102_PCA256_ROOTSIFT (128 -> 64)
  * fisher_vgg16_pca    GMMWeights.OXFORD102_K256_VGG16_PCA + _PCA.OXFORD102_PCA256_VGG16 (514 -> 257), n = 196 rows per image:
                        the reference's own shape for BASELINE configs[2] (examples/pipeline.ipynb: FV length 131,840)
  * vlad_rootsift       the K = 256 RootSIFT codebook derived from the mixture's means_ (the reference's KMeans file is absent)

Inputs are seeded synthetic descriptors (pvsim.synth); the deep-feature rows are samples of the mixture mapped back through the
PCA (x = z . components + mean + noise), so that the posteriors are as peaked as on real features.  Stored: the inputs needed
to regenerate nothing else, the reference's outputs (first image whole, every 16th element of all images), labels.
"""
from __future__ import annotations

import os
import sys
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402

MODELS = os.path.join(mg.REPO, "python-visual-similarity_amd", "pvsim", "res", "model_files")


from shipped_inputs import shipped_inputs  # noqa: E402  (shared with the tests)


def _gmm(name):
    z = np.load(os.path.join(MODELS, name + ".npz"), allow_pickle=False)
    g = mg._gmm_with_tables(z["weights"], z["means"], z["covariances"])
    g.precisions_cholesky_ = np.array(z["precisions_cholesky"])        # as stored in the reference's file
    g.precisions_ = g.precisions_cholesky_ ** 2
    return g


def _pca(name):
    z = np.load(os.path.join(MODELS, name + ".npz"), allow_pickle=False)
    p = mg._pca_with_tables(z["components"], z["mean"])
    p.explained_variance_ = np.array(z["explained_variance"])
    return p


def main():
    check = "--check" in sys.argv
    from pvsim import synth
    VLADEncoder, FisherVectorEncoder, Pipeline, Lambda, ref_cos, ref_eval = mg._import_reference()
    warnings.simplefilter("ignore")
    raws, deep = shipped_inputs()
    rootsift_x = Lambda(lambda im: synth.rootsift(im.astype(np.float32)), 128)
    images = [r.astype(np.int64) for r in raws]

    out = {}
    g = _gmm("gmm_k256_root_sift_no_pca")
    F = FisherVectorEncoder(feature_extractor=rootsift_x, gmm_model=g).encode(images)
    assert F.shape == (5, 256 + 2 * 256 * 128) and F.dtype == np.float64
    out.update(fisher_rootsift_img0=F[0], fisher_rootsift_every16=np.ascontiguousarray(F[:, ::16]),
               fisher_rootsift_cos=ref_cos(F, F), resp_rootsift_img1=g.predict_proba(synth.rootsift(raws[1].astype(np.float32))))

    gp, pp = _gmm("gmm_k256_root_sift_pca"), _pca("pca_k256_root_sift_f2")
    Fp = FisherVectorEncoder(feature_extractor=rootsift_x, gmm_model=gp, pca=pp).encode(images)
    assert Fp.shape == (5, 256 + 2 * 256 * 64)
    out.update(fisher_rootsift_pca_img0=Fp[0], fisher_rootsift_pca_every16=np.ascontiguousarray(Fp[:, ::16]))

    gd, pd_ = _gmm("gmm_k256_deep_features_vgg16_pca"), _pca("pca_k256_deep_features_vgg16_f2")
    store = {i: d for i, d in enumerate(deep)}
    deep_x = Lambda(lambda im: store[int(im[0, 0])], 514)
    Fd = FisherVectorEncoder(feature_extractor=deep_x, gmm_model=gd, pca=pd_).encode([np.array([[i]], dtype=np.int64) for i in range(len(deep))])
    assert Fd.shape == (4, 131840) and Fd.dtype == np.float64          # examples/pipeline.ipynb cell 12: (1, 131840)
    out.update(fisher_vgg16_pca_img0=Fd[0], fisher_vgg16_pca_every16=np.ascontiguousarray(Fd[:, ::16]), fisher_vgg16_pca_cos=ref_cos(Fd, Fd))

    C = np.ascontiguousarray(np.load(os.path.join(MODELS, "k_means_k256_root_sift_no_pca.npz"), allow_pickle=False)["cluster_centers"])
    km = mg._kmeans_with_centres(C)
    V = VLADEncoder(feature_extractor=rootsift_x, kmeans_model=km).encode(images)
    labels = np.concatenate([km.predict(synth.rootsift(r.astype(np.float32))) for r in raws]).astype(np.int32)
    out.update(vlad_rootsift=V, vlad_labels=labels, vlad_cos=ref_cos(V, V))

    path = os.path.join(HERE, "shipped_tables.npz")
    if check:
        old = np.load(path, allow_pickle=False)
        bad = [k for k, v in out.items() if k not in old.files or np.ascontiguousarray(old[k]).tobytes() != np.ascontiguousarray(v).tobytes()]
        print("shipped_tables.npz:", "reproduced bit for bit" if not bad else f"DIFFERS in {bad}")
        sys.exit(1 if bad else 0)
    np.savez_compressed(path, **out)
    print(f"wrote shipped_tables.npz ({os.path.getsize(path) / 1e6:.2f} MB):", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
