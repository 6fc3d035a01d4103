#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Run in the build container only (it needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py            # (re)write the fixtures
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --check    # run the reference again, assert bit-equality
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --refit    # fit new tables (changes every fixture)

The codebook / GMM / PCA tables are read from the committed tables_k256_d128.npz when it exists (KMeans' OpenMP chunk sums
make a re-fit differ in the last bits from run to run), so a fresh run reproduces the committed files bit for bit.

What it does
  * imports pyvisim's own VLADEncoder / FisherVectorEncoder / Pipeline / cosine_similarity / eval
    from /root/reference, after registering inert stand-ins for the four modules that are absent
    from this image (cv2, h5py, seaborn, torchvision) -- recipe of SURVEY.md section 8(c);
  * never writes under /root/reference: byte-code writing is off and the package's import-time
    `os.makedirs(<root>/res/logs)` (pyvisim/_config.py:9-11) is turned into a no-op for that path;
  * never unpickles the reference's *.pkl model files (they are joblib pickles; this run's rules
    allow only loaders that execute nothing).  Codebooks / GMM / PCA tables are instead FIT HERE
    with scikit-learn on seeded synthetic descriptors and handed to the reference encoders as
    live sklearn objects;
  * drives the encoders at descriptor level through the reference's own `Lambda` extractor:
    the "image" is a 2-D integer-valued SIFT-like array (passes is_numpy_image, _utils.py:34-53).

Outputs are DATA ONLY (inputs + the reference's outputs) as .npz files, plus versions.json.
"""
from __future__ import annotations

import json
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REPO, "python-visual-similarity_amd"))

import numpy as np  # noqa: E402


def _install_standins() -> None:
    import torch

    cv2 = types.ModuleType("cv2")

    class _SIFT:
        @staticmethod
        def create():
            raise RuntimeError("cv2 stand-in: SIFT is not available offline")

    cv2.SIFT = _SIFT
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda *a, **k: None
    cv2.cvtColor = lambda img, code: img
    sys.modules["cv2"] = cv2

    for name in ("h5py", "seaborn"):
        sys.modules[name] = types.ModuleType(name)

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    trf = types.ModuleType("torchvision.transforms.functional")
    tm = types.ModuleType("torchvision.models")

    class Compose:  # annotation target only
        def __init__(self, ts):
            self.ts = ts

    tr.Compose = Compose
    tr.ToTensor = lambda: None
    tr.Resize = lambda *a, **k: None
    tr.functional = trf

    class VGG16_Weights:
        DEFAULT = None

    def vgg16(weights=None):
        # evaluated as a default argument at import (pyvisim/features/_features.py:179);
        # the real one would download weights -- never attempted.
        return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3))

    tm.vgg16 = vgg16
    tm.VGG16_Weights = VGG16_Weights
    tv.transforms = tr
    tv.models = tm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr,
                        "torchvision.transforms.functional": trf, "torchvision.models": tm})


def _import_reference():
    _install_standins()
    real_makedirs = os.makedirs

    def guarded_makedirs(path, *a, **k):
        if os.path.abspath(str(path)).startswith(REF):
            return None  # the reference tree is read-only by rule
        return real_makedirs(path, *a, **k)

    os.makedirs = guarded_makedirs
    sys.path.insert(0, REF)
    try:
        import io
        from contextlib import redirect_stdout
        with redirect_stdout(io.StringIO()):  # logging-config complaint about the missing log dir
            from pyvisim.encoders import VLADEncoder, FisherVectorEncoder, Pipeline
            from pyvisim.features import Lambda
            from pyvisim._utils import cosine_similarity
            from pyvisim import eval as ref_eval
    finally:
        os.makedirs = real_makedirs
    assert not os.path.exists(os.path.join(REF, "res", "logs")), "reference tree was written to"
    return VLADEncoder, FisherVectorEncoder, Pipeline, Lambda, cosine_similarity, ref_eval


def _kmeans_with_centres(C, seed=0):
    """A fitted KMeans whose centres are exactly C (one-iteration fit, then overwrite)."""
    from sklearn.cluster import KMeans
    C = np.ascontiguousarray(C, dtype=np.float32)
    km = KMeans(n_clusters=C.shape[0], init=C, n_init=1, max_iter=1, random_state=seed).fit(C)
    km.cluster_centers_ = C.copy()
    return km


def _gmm_with_tables(w, mu, cov):
    from sklearn.mixture import GaussianMixture
    g = GaussianMixture(n_components=len(w), covariance_type="diag")
    g.weights_, g.means_, g.covariances_ = np.asarray(w, np.float64), np.asarray(mu, np.float64), np.asarray(cov, np.float64)
    g.precisions_cholesky_ = 1.0 / np.sqrt(g.covariances_)       # sklearn _compute_precision_cholesky, 'diag'
    g.precisions_ = g.precisions_cholesky_ ** 2
    g.converged_, g.n_iter_, g.lower_bound_ = True, 1, 0.0
    g.n_features_in_ = g.means_.shape[1]
    return g


def _pca_with_tables(components, mean):
    from sklearn.decomposition import PCA
    p = PCA(n_components=components.shape[0])
    # a fitted PCA keeps components_ as the transposed (Fortran-ordered) eigenvector block; the GEMM of transform() takes a
    # different BLAS path for a C-ordered copy, so the layout is part of reproducing the fitted object bit for bit
    p.components_, p.mean_ = np.asfortranarray(components), np.asarray(mean)
    p.n_components_, p.n_features_in_, p.whiten = components.shape[0], components.shape[1], False
    p.explained_variance_ = np.ones(components.shape[0], components.dtype)
    return p


sys.path.insert(0, HERE)
from config3_inputs import config3_inputs  # noqa: E402  (shared with the tests)


def main() -> None:
    check = "--check" in sys.argv
    refit = "--refit" in sys.argv
    from sklearn.cluster import KMeans
    from sklearn.decomposition import PCA
    from sklearn.mixture import GaussianMixture
    import sklearn, scipy, joblib
    from pvsim import synth

    VLADEncoder, FisherVectorEncoder, Pipeline, Lambda, ref_cos, ref_eval = _import_reference()
    warnings.simplefilter("ignore")
    out = {}

    bad = []

    def save(name, **arrays):
        path = os.path.join(HERE, name + ".npz")
        out[name] = {k: [list(np.shape(v)), str(np.asarray(v).dtype)] for k, v in arrays.items()}
        if not os.path.exists(path):
            if check:
                bad.append(f"{name}.npz (missing)")
                return
        else:
            old = np.load(path, allow_pickle=False)
            n_bad = len(bad)
            for k, v in arrays.items():
                v = np.asarray(v)
                same = k in old.files and old[k].shape == v.shape and old[k].dtype == v.dtype and \
                    np.ascontiguousarray(old[k]).tobytes() == np.ascontiguousarray(v).tobytes()
                if not same:
                    bad.append(f"{name}.npz:{k}")
            same_file = len(bad) == n_bad and sorted(old.files) == sorted(arrays)
            if check:
                print(f"  checked {name}.npz: {'ok' if same_file else 'DIFFERS'}")
                return
            del bad[n_bad:]
            if same_file:
                print(f"  {name}.npz unchanged")       # not rewritten: a zip archive carries time stamps
                return
        np.savez_compressed(path, **arrays)
        print(f"  wrote {name}.npz  ({os.path.getsize(path)/1e6:.2f} MB)")

    proto = synth.sift_prototypes()
    rootsift_x = Lambda(synth.rootsift, 128)

    # ---------------------------------------------------------------- tables (fit here, seeded)
    tab_path = os.path.join(HERE, "tables_k256_d128.npz")
    if os.path.exists(tab_path) and not refit:
        T = np.load(tab_path, allow_pickle=False)
        C = np.ascontiguousarray(T["centroids"], dtype=np.float32)
        km = _kmeans_with_centres(C)
        gmm = _gmm_with_tables(T["gmm_weights"], T["gmm_means"], T["gmm_covariances"])
        pca = _pca_with_tables(T["pca_components"], T["pca_mean"])
        km_p = _kmeans_with_centres(T["centroids_pca64"], 1)
        gmm_p = _gmm_with_tables(T["gmmp_weights"], T["gmmp_means"], T["gmmp_covariances"])
        print("  tables: loaded from the committed tables_k256_d128.npz")
    else:
        rng = np.random.default_rng(20240)
        train = synth.rootsift(synth.sift_like(60000, rng, proto))
        km = KMeans(n_clusters=256, n_init=1, max_iter=25, random_state=0).fit(train)
        C = np.ascontiguousarray(km.cluster_centers_, dtype=np.float32)
        km.cluster_centers_ = C.copy()
        gmm = GaussianMixture(n_components=256, covariance_type="diag", max_iter=15, random_state=0,
                              init_params="kmeans").fit(train[:40000].astype(np.float64))
        pca = PCA(n_components=64, random_state=0).fit(train[:20000])
        trainp = pca.transform(train[:40000])
        km_p = KMeans(n_clusters=256, n_init=1, max_iter=15, random_state=1).fit(trainp)
        km_p.cluster_centers_ = np.ascontiguousarray(km_p.cluster_centers_, dtype=np.float32)
        gmm_p = GaussianMixture(n_components=64, covariance_type="diag", max_iter=10,
                                random_state=1).fit(trainp.astype(np.float64))
    save("tables_k256_d128",
         centroids=C,
         gmm_weights=gmm.weights_, gmm_means=gmm.means_, gmm_covariances=gmm.covariances_,
         pca_components=np.ascontiguousarray(pca.components_), pca_mean=pca.mean_,
         centroids_pca64=km_p.cluster_centers_,
         gmmp_weights=gmm_p.weights_, gmmp_means=gmm_p.means_, gmmp_covariances=gmm_p.covariances_)
    print("  covariance entries at the reg_covar floor:", int((gmm.covariances_ <= 1.0000001e-6).sum()))

    # ---------------------------------------------------------------- G2: VLAD, K=256 D=128
    rng = np.random.default_rng(1234)
    counts = [1, 7, 64, 200, 512, 513, 1257, 2100]
    raws = [synth.sift_like(n, rng, proto) for n in counts]
    enc = VLADEncoder(feature_extractor=rootsift_x, kmeans_model=km)
    V = enc.encode(raws)
    labels = np.concatenate([km.predict(synth.rootsift(r)) for r in raws]).astype(np.int32)
    assert V.shape == (len(counts), 256 * 128) and V.dtype == np.float32
    save("vlad_k256_d128",
         raw_u8=np.concatenate(raws).astype(np.uint8), offsets=np.cumsum([0] + counts).astype(np.int64),
         labels=labels, vlad=V)

    # VLAD with PCA(128->64) prologue (vlad.py:89-90)
    enc_p = VLADEncoder(feature_extractor=rootsift_x, kmeans_model=km_p, pca=pca)
    Vp = enc_p.encode(raws[:6])
    save("vlad_pca64_k256", n_images=np.int64(6), vlad=Vp)

    # ---------------------------------------------------------------- G5: small K/D, parameter variants
    rng = np.random.default_rng(5)
    Ks, Ds = 16, 8
    small_train = rng.random((4000, Ds)).astype(np.float32)
    small_path = os.path.join(HERE, "small_k16_d8.npz")
    if os.path.exists(small_path) and not refit:
        S_ = np.load(small_path, allow_pickle=False)
        Cs = np.ascontiguousarray(S_["centroids"], dtype=np.float32)
        km_s = _kmeans_with_centres(Cs, 2)
        gmm_s = _gmm_with_tables(S_["gmm_weights"], S_["gmm_means"], S_["gmm_covariances"])
    else:
        km_s = KMeans(n_clusters=Ks, n_init=1, max_iter=10, random_state=2).fit(small_train)
        Cs = np.ascontiguousarray(km_s.cluster_centers_, dtype=np.float32)
        Cs[5] = Cs[3]           # duplicated centroid -> exact tie, first index must win (G6)
        km_s.cluster_centers_ = Cs.copy()
        gmm_s = GaussianMixture(n_components=Ks, covariance_type="diag", max_iter=10,
                                random_state=2).fit(small_train.astype(np.float64))
    ident = Lambda(lambda im: im.astype(np.float32) / np.float32(16.0), Ds)
    small_raw = [rng.integers(0, 17, size=(n, Ds)).astype(np.float32) for n in (1, 3, 40, 40, 333)]
    small_raw[3] = small_raw[2].copy()  # duplicate image -> tied similarity rows
    variants = {}
    for tag, kw in {"default": {}, "p05": {"power_norm_weight": 0.5}, "l1": {"norm_order": 1},
                    "p03_l1": {"power_norm_weight": 0.3, "norm_order": 1},
                    "eps": {"epsilon": 1e-3}}.items():
        variants["vlad_" + tag] = VLADEncoder(feature_extractor=ident, kmeans_model=km_s, **kw).encode(small_raw)
    for tag, kw in {"default": {}, "p1": {"power_norm_weight": 1.0}, "l1": {"norm_order": 1},
                    "p03": {"power_norm_weight": 0.3}}.items():
        variants["fisher_" + tag] = FisherVectorEncoder(feature_extractor=ident, gmm_model=gmm_s, **kw).encode(small_raw)
    small_lab = np.concatenate([km_s.predict(r.astype(np.float32) / np.float32(16.0)) for r in small_raw]).astype(np.int32)
    # the reference's empty-image quirk (vlad.py:92-93): whole batch collapses to one zero vector
    quirk = VLADEncoder(feature_extractor=ident, kmeans_model=km_s).encode(
        [small_raw[0], np.zeros((0, Ds), np.float32), small_raw[1]])
    save("small_k16_d8",
         raw=np.concatenate(small_raw), offsets=np.cumsum([0] + [len(r) for r in small_raw]).astype(np.int64),
         centroids=Cs, labels=small_lab,
         gmm_weights=gmm_s.weights_, gmm_means=gmm_s.means_, gmm_covariances=gmm_s.covariances_,
         empty_quirk=quirk, **variants)

    # ---------------------------------------------------------------- Fisher, K=256 D=128 (fp64 out)
    fenc = FisherVectorEncoder(feature_extractor=rootsift_x, gmm_model=gmm)
    sel = [1, 3, 4, 6]  # n = 7, 200, 512, 1257
    F = fenc.encode([raws[i] for i in sel])
    assert F.shape == (4, 256 + 2 * 256 * 128) and F.dtype == np.float64
    resp = gmm.predict_proba(synth.rootsift(raws[3]))
    save("fisher_k256_d128", image_index=np.asarray(sel, np.int64), fisher=F, resp_img3=resp)

    # Fisher with PCA(128->64), K=64
    fenc_p = FisherVectorEncoder(feature_extractor=rootsift_x, gmm_model=gmm_p, pca=pca)
    Fp = fenc_p.encode(raws[:6])
    save("fisher_pca64_k64", n_images=np.int64(6), fisher=Fp)

    # Fisher on deep-feature-like rows (D=96, K=32, n=196) -- config-3 shaped, small
    rng = np.random.default_rng(1236)
    centers = rng.normal(0.0, 2.0, size=(24, 96))
    dtrain = synth.deep_like(20000, 96, rng, centers)
    gmm_d = GaussianMixture(n_components=32, covariance_type="diag", max_iter=10,
                            random_state=3).fit(dtrain.astype(np.float64))
    deep_imgs = [synth.deep_like(196, 96, rng, centers) for _ in range(5)]
    store = {}
    deep_x = Lambda(lambda im: store[int(im[0, 0])], 96)
    for i, d in enumerate(deep_imgs):
        store[i] = d
    Fd = FisherVectorEncoder(feature_extractor=deep_x, gmm_model=gmm_d).encode(
        [np.array([[i]], dtype=np.int64) for i in range(5)])
    save("fisher_deep_k32_d96", desc=np.stack(deep_imgs), fisher=Fd,
         gmm_weights=gmm_d.weights_, gmm_means=gmm_d.means_, gmm_covariances=gmm_d.covariances_)

    # ---------------------------------------------------------------- G3: cosine / similarity_score
    rng = np.random.default_rng(99)
    a32 = rng.normal(size=(5, 40)).astype(np.float32)
    b32 = rng.normal(size=(10, 40)).astype(np.float32)
    b32[2] = 0.0                 # zero row -> stays zero (sklearn normalize: zero norms -> 1)
    b32[7] = b32[4]              # duplicate row -> tied scores
    a64, b64 = a32.astype(np.float64) * 1.7, b32.astype(np.float64)
    v64x64 = ref_cos(V, V)
    simscore = enc.similarity_score(raws[:3], raws[2:7])
    save("cosine",
         a32=a32, b32=b32, cos32=ref_cos(a32, b32), cos64=ref_cos(a64, b64),
         cos_mixed=ref_cos(a32, b64), cos_1d=ref_cos(a32[0], b32[1]),
         vlad_self=v64x64, similarity_score_3x5=simscore)

    # Pipeline (pipeline.py:47-66): hstack of flattened encodings
    pipe = Pipeline([enc, fenc])
    P = pipe.encode(raws[1:4])
    save("pipeline", encoded=P, score=pipe.similarity_score(raws[1:3], raws[2:4]))

    # ---------------------------------------------------------------- G4: retrieval / eval
    rng = np.random.default_rng(4321)
    n_db, n_q, n_cls = 64, 12, 6
    cls_proto = [proto[rng.choice(len(proto), 24, replace=False)] for _ in range(n_cls)]
    db_lab = rng.integers(0, n_cls, size=n_db)
    q_lab = rng.integers(0, n_cls, size=n_q)
    db_raw = [synth.sift_like(int(rng.integers(60, 300)), rng, cls_proto[l]) for l in db_lab]
    q_raw = [synth.sift_like(int(rng.integers(60, 300)), rng, cls_proto[l]) for l in q_lab]
    paths = [f"img_{i:03d}.jpg" for i in range(n_db)]
    db_vecs = enc.encode(db_raw)
    encoding_map = dict(zip(paths, db_vecs))
    path_labels = dict(zip(paths, [int(l) for l in db_lab]))
    top = [ref_eval.retrieve_top_k_similar([q], encoding_map, enc, k=7) for q in q_raw]
    # (queries are 2-D "images": wrap in a list so encode() sees one image)
    wrapped = [[q] for q in q_raw]
    res = {
        "acc_k1": ref_eval.top_k_accuracy(wrapped, list(q_lab), encoding_map, path_labels, enc, 1),
        "acc_k5": ref_eval.top_k_accuracy(wrapped, list(q_lab), encoding_map, path_labels, enc, 5),
        "map_all": ref_eval.top_k_map(wrapped, list(q_lab), encoding_map, path_labels, enc, None),
        "map_k5": ref_eval.top_k_map(wrapped, list(q_lab), encoding_map, path_labels, enc, 5),
        "map_k10": ref_eval.top_k_map(wrapped, list(q_lab), encoding_map, path_labels, enc, 10),
    }
    q_vecs = enc.encode(q_raw)
    sims = ref_cos(q_vecs, db_vecs)
    save("eval_db64",
         db_raw_u8=np.concatenate(db_raw).astype(np.uint8),
         db_offsets=np.cumsum([0] + [len(r) for r in db_raw]).astype(np.int64),
         q_raw_u8=np.concatenate(q_raw).astype(np.uint8),
         q_offsets=np.cumsum([0] + [len(r) for r in q_raw]).astype(np.int64),
         db_labels=db_lab.astype(np.int64), q_labels=q_lab.astype(np.int64),
         top7_index=np.asarray([[paths.index(p) for p, _ in t] for t in top], np.int64),
         top7_score=np.asarray([[s for _, s in t] for t in top], np.float32),
         sims=sims, full_argsort=np.argsort(-sims, axis=1).astype(np.int64),
         **{k: np.float64(v) for k, v in res.items()})

    # ---------------------------------------------------------------- eval.* with a FisherVectorEncoder: float64 scores, float64 ranking
    fdb = fenc.encode(db_raw)
    assert fdb.dtype == np.float64
    f_map = dict(zip(paths, fdb))
    f_top = [ref_eval.retrieve_top_k_similar([q], f_map, fenc, k=7) for q in q_raw]
    f_res = {
        "acc_k1": ref_eval.top_k_accuracy(wrapped, list(q_lab), f_map, path_labels, fenc, 1),
        "acc_k5": ref_eval.top_k_accuracy(wrapped, list(q_lab), f_map, path_labels, fenc, 5),
        "map_all": ref_eval.top_k_map(wrapped, list(q_lab), f_map, path_labels, fenc, None),
        "map_k5": ref_eval.top_k_map(wrapped, list(q_lab), f_map, path_labels, fenc, 5),
    }
    f_sims = ref_cos(fenc.encode(q_raw), fdb)
    assert f_sims.dtype == np.float64 and all(isinstance(sc, np.float64) for _, sc in f_top[0])
    save("eval_fisher_db64",
         top7_index=np.asarray([[paths.index(p) for p, _ in t] for t in f_top], np.int64),
         top7_score=np.asarray([[sc for _, sc in t] for t in f_top], np.float64),
         sims=f_sims, full_argsort=np.argsort(-f_sims, axis=1).astype(np.int64),
         **{k: np.float64(v) for k, v in f_res.items()})

    # ---------------------------------------------------------------- BASELINE configs[2] shape: Fisher D = 512, K = 256, n = 196
    # inputs and tables are regenerated from the seed wherever they are needed (config3_inputs); the fixture keeps the
    # reference's outputs only: image 0 whole, every 32nd element of all 12 images, and the row norms
    w3, mu3, cov3, imgs3 = config3_inputs(12)
    store3 = {i: d for i, d in enumerate(imgs3)}
    x3 = Lambda(lambda im: store3[int(im[0, 0])], 512)
    F3 = FisherVectorEncoder(feature_extractor=x3, gmm_model=_gmm_with_tables(w3, mu3, cov3)).encode(
        [np.array([[i]], dtype=np.int64) for i in range(12)])
    assert F3.shape == (12, 256 + 2 * 256 * 512) and F3.dtype == np.float64
    save("fisher_k256_d512", image0=F3[0], every32=np.ascontiguousarray(F3[:, ::32]), cos_self=ref_cos(F3, F3),
         desc_checksum=np.float64(sum(float(d.astype(np.float64).sum()) for d in imgs3)))

    if check:
        if bad:
            print("DIFFERENT from the committed fixtures:", *bad, sep="\n  ")
            sys.exit(1)
        print("all fixtures reproduced bit for bit")
        return
    meta = {"numpy": np.__version__, "scikit-learn": sklearn.__version__, "scipy": scipy.__version__,
            "joblib": joblib.__version__, "python": sys.version.split()[0],
            "reference": "MechaCritter/Python-Visual-Similarity pyvisim 0.1.3 (/root/reference)",
            "files": out}
    with open(os.path.join(HERE, "versions.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
