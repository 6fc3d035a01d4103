"""CPU tests of the host side: the C-ABI library loads and exports every symbol of include/pvsim.h, the
drop-in classes validate like the reference's, and nothing computes without a GPU (fails loudly)."""
import os
import re
import warnings

import numpy as np
import pytest

import pvsim_oracle as orc
from conftest import REPO, load_golden


@pytest.fixture(scope="session", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()


def _header_symbols():
    """every function declared in include/*.h: pvsim.h (the product ABI) and pvsim_diag.h (diagnostics)"""
    syms = set()
    for name in sorted(os.listdir(os.path.join(REPO, "include"))):
        if not name.endswith(".h"):
            continue
        text = open(os.path.join(REPO, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(pvs_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_library_exports_every_header_symbol():
    import ctypes
    from pvsim import _ffi
    syms = _header_symbols()
    assert len(syms) >= 35
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/*.h but not exported"
    assert sorted(_ffi.SIGNATURES) == syms            # the ctypes table binds exactly the header
    assert _ffi.lib().pvs_version() == 103


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pvsim
    from pvsim._utils import cosine_similarity
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pvsim.Context(0)
    with pytest.raises(RuntimeError):
        cosine_similarity(np.ones((2, 4), np.float32), np.ones((3, 4), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "python-visual-similarity_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                assert "oracle" not in open(os.path.join(root, f)).read().lower(), f


def _fx(dim):
    from pvsim.features import Lambda
    return Lambda(lambda im: im.astype(np.float32), dim)


def test_constructor_validation_matches_reference():
    from pvsim.encoders import VLADEncoder, FisherVectorEncoder, Pipeline, KMeansWeights, GMMWeights
    from pvsim.models import KMeansModel, GMMModel, PCAModel
    km = KMeansModel(np.random.rand(16, 8))
    gm = GMMModel(np.full(4, 0.25), np.random.rand(4, 8), np.ones((4, 8)))
    with pytest.raises(TypeError):                         # _base_encoder.py:228-231
        VLADEncoder(feature_extractor="sift", kmeans_model=km)
    with pytest.raises(ValueError):                        # vlad.py:55-59
        VLADEncoder(_fx(8), kmeans_model=gm)
    with pytest.raises(ValueError):
        FisherVectorEncoder(_fx(8), gmm_model=km)
    with pytest.raises(ValueError):
        VLADEncoder(_fx(8), weights=GMMWeights.OXFORD102_K256_ROOTSIFT)
    with pytest.raises(RuntimeError):                      # extractor dim != model dim
        VLADEncoder(_fx(10), kmeans_model=km)
    pca_bad = PCAModel(np.random.rand(4, 10), np.zeros(10))
    with pytest.raises(ValueError):                        # PCA input != extractor output
        VLADEncoder(_fx(8), kmeans_model=km, pca=pca_bad)
    pca = PCAModel(np.random.rand(4, 8), np.zeros(8))
    with pytest.warns(UserWarning):                        # incompatible PCA is reset (default: no raise)
        e = VLADEncoder(_fx(8), kmeans_model=KMeansModel(np.random.rand(16, 8)), pca=pca)
    assert e.pca is None
    with pytest.raises(RuntimeError):
        VLADEncoder(_fx(8), kmeans_model=KMeansModel(np.random.rand(16, 8)), pca=pca,
                    raise_error_when_pca_incompatible=True)
    ok = VLADEncoder(_fx(8), kmeans_model=KMeansModel(np.random.rand(16, 4)), pca=pca)
    assert ok.pca is pca and ok.power_norm_weight == 1 and ok.norm_order == 2 and ok.epsilon == 1e-9
    assert FisherVectorEncoder(_fx(8), gmm_model=gm).power_norm_weight == 0.5
    with pytest.raises(ValueError):
        Pipeline([ok, "not an encoder"])
    with pytest.raises(FileNotFoundError):                 # a table the reference's checkout does not hold either is reported
        GMMWeights.OXFORD102_K256_VGG16.load()
    g = GMMWeights.OXFORD102_K256_ROOTSIFT.load()          # the reference's shipped tables, as plain arrays (never unpickled)
    assert g.means_.shape == (256, 128) and g.means_.dtype == np.float64 and int((g.covariances_ < 1.01e-6).sum()) == 487
    with pytest.warns(UserWarning, match="absent from its checkout"):     # KMeans files are missing upstream: derived stand-in
        km2 = KMeansWeights.OXFORD102_K256_ROOTSIFT.load()
    assert km2.cluster_centers_.shape == (256, 128) and km2.cluster_centers_.dtype == np.float32
    from pvsim.encoders._base_encoder import _PCA
    assert _PCA.OXFORD102_PCA256_VGG16.load().components_.shape == (257, 514)
    assert [m.name for m in KMeansWeights] == [m.name for m in GMMWeights]
    assert "VLADEncoder(feature_extractor=Lambda" in repr(ok)


def test_sklearn_objects_are_accepted():
    sk = pytest.importorskip("sklearn.cluster")
    from sklearn.mixture import GaussianMixture
    from pvsim.encoders import VLADEncoder, FisherVectorEncoder
    x = np.random.default_rng(0).random((200, 8)).astype(np.float32)
    km = sk.KMeans(n_clusters=4, n_init=1, random_state=0).fit(x)
    gm = GaussianMixture(4, covariance_type="diag", random_state=0).fit(x)
    assert VLADEncoder(_fx(8), kmeans_model=km).clustering_model is km
    assert FisherVectorEncoder(_fx(8), gmm_model=gm).clustering_model is gm
    with pytest.raises(ValueError):
        VLADEncoder(_fx(8), kmeans_model=gm)


def test_input_validation_and_torch_rejection():
    import torch
    from pvsim._errors import InvalidImageError
    from pvsim.encoders import VLADEncoder
    from pvsim.features import Lambda, RootSIFT, SIFT
    from pvsim.models import KMeansModel
    e = VLADEncoder(_fx(8), kmeans_model=KMeansModel(np.random.rand(16, 8)))
    with pytest.raises(RuntimeError, match="Torch images"):
        e.encode(torch.zeros(3, 8, 8))
    with pytest.raises(InvalidImageError):
        e.encode([np.full((4, 8), 0.5)])                   # 2-D non-integer "mask"
    with pytest.raises(InvalidImageError):
        e.encode(np.zeros((4, 4, 2)))                      # 3-D but not (H, W, 3)
    with pytest.raises(ValueError):
        Lambda("nope", 3)
    bad = Lambda(lambda im: np.zeros((2, 5), np.float32), 8)
    with pytest.raises(ValueError):
        bad(np.zeros((2, 2), np.int64))
    assert Lambda(lambda im: None, 8)(np.zeros((2, 2), np.int64)).shape == (0, 8)
    assert RootSIFT().output_dim == 128 and SIFT().output_dim == 128
    with pytest.raises(ImportError):                       # cv2 absent: fails when called, not at import
        RootSIFT()(np.zeros((8, 8, 3), np.uint8))


def test_custom_similarity_function_contract():
    from pvsim.encoders import VLADEncoder
    from pvsim.encoders._base_encoder import check_desired_output
    from pvsim.models import KMeansModel
    km = KMeansModel(np.random.rand(16, 8))
    batch = lambda a, b: a @ b.T
    assert VLADEncoder(_fx(8), kmeans_model=km, similarity_func=batch).similarity_func is batch
    scalar = lambda a, b: float((a * b).sum())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        f = check_desired_output(scalar, np.random.rand(10, 10), np.random.rand(10, 10))
    assert w and f is not scalar
    a, b = np.random.rand(3, 5), np.random.rand(4, 5)
    out = f(a, b)
    assert out.shape == (3, 4) and out.dtype == np.float32
    np.testing.assert_allclose(out, a @ b.T, rtol=1e-6)
    with pytest.warns(UserWarning):
        g = check_desired_output(lambda a, b: 1 / 0, np.random.rand(10, 10), np.random.rand(10, 10))
    assert g is not None


def test_pack_descriptors_and_models_roundtrip(tmp_path):
    from pvsim import pack_descriptors
    from pvsim.models import KMeansModel, GMMModel, PCAModel, save_model, load_model
    packed, off = pack_descriptors([np.ones((2, 3)), np.zeros((0, 3)), None, 2 * np.ones((1, 3))], 3)
    assert list(off) == [0, 2, 2, 2, 3] and packed.dtype == np.float32 and packed[2, 0] == 2
    with pytest.raises(RuntimeError):
        pack_descriptors([np.ones((2, 4))], 3)
    for m in (KMeansModel(np.random.rand(5, 3)), GMMModel(np.full(2, .5), np.random.rand(2, 3), np.ones((2, 3))),
              PCAModel(np.random.rand(2, 3), np.random.rand(3))):
        p = str(tmp_path / (type(m).__name__ + ".npz"))
        save_model(p, m)
        r = load_model(p)
        assert type(r) is type(m) and r.n_features_in_ == m.n_features_in_


def test_eval_bookkeeping_against_golden(monkeypatch, tables):
    """The host half of eval.py (label bookkeeping, AP with in-window R, accuracy denominator) with the
    device ranking replaced by the oracle's -- checks the Python logic on CPU against the reference's values."""
    from pvsim import eval as ev
    from pvsim import synth
    g = load_golden("eval_db64")
    C = tables["centroids"]
    db = [synth.rootsift(r) for r in orc.split_ragged(g["db_raw_u8"].astype(np.float32), g["db_offsets"])]
    qs = [synth.rootsift(r) for r in orc.split_ragged(g["q_raw_u8"].astype(np.float32), g["q_offsets"])]
    dbv, qv = orc.vlad_encode(db, C), orc.vlad_encode(qs, C)

    class FakeEncoder:
        def __init__(self):
            self.i = 0

        def encode(self, image):
            v = qv[self.i % len(qv)][None]
            self.i += 1
            return v

    def fake_rank(query_vecs, all_vectors, k, ctx=None, resident=None):
        kk = all_vectors.shape[0] if k is None else min(k, all_vectors.shape[0])
        return orc.topk(orc.cosine_similarity(query_vecs, all_vectors), kk)

    monkeypatch.setattr(ev, "_rank", fake_rank)
    paths = [f"img_{i:03d}.jpg" for i in range(len(dbv))]
    emap, plab = dict(zip(paths, dbv)), dict(zip(paths, [int(l) for l in g["db_labels"]]))
    imgs, labels = [[q] for q in qs], list(g["q_labels"])
    assert ev.top_k_accuracy(imgs, labels, emap, plab, FakeEncoder(), 1) == float(g["acc_k1"])
    assert ev.top_k_accuracy(imgs, labels, emap, plab, FakeEncoder(), 5) == float(g["acc_k5"])
    for k, key in ((None, "map_all"), (5, "map_k5"), (10, "map_k10")):
        assert abs(ev.top_k_map(imgs, labels, emap, plab, FakeEncoder(), k) - float(g[key])) < 1e-12
    top = ev.retrieve_top_k_similar(imgs[0], emap, FakeEncoder(), k=7)
    assert [paths.index(p) for p, _ in top] == list(g["top7_index"][0])


def test_index_persistence_roundtrip(tmp_path):
    from pvsim import index
    rng = np.random.default_rng(0)
    emap = {f"b/{i}.jpg": rng.random(6).astype(np.float32) for i in (3, 1, 2)}
    p = str(tmp_path / "idx.npz")
    index.save_encoding_map(p, emap)
    back = index.load_encoding_map(p)
    assert list(back) == list(emap) and all(np.array_equal(back[k], emap[k]) for k in emap)   # insertion order kept
    v = rng.random((5, 4)).astype(np.float32)
    index.save_shard(str(tmp_path / "sh"), 1, 2, 3, v[3:], ["d", "e"])
    index.save_shard(str(tmp_path / "sh"), 0, 2, 0, v[:3], ["a", "b", "c"])
    vecs, paths = index.load_shards(str(tmp_path / "sh"))
    assert np.array_equal(vecs, v) and paths == list("abcde")


@pytest.mark.parametrize("order", ["engine_first", "torch_first"])
def test_one_hip_runtime_is_mapped_in_either_import_order(order):
    """libpvsim_hip.so needs libamdhip64.so.7; a PyTorch-ROCm wheel bundles its own copy with that SONAME.  Whatever is
    imported first, exactly ONE runtime image may end up mapped (pvsim/_ffi.py:_one_hip_runtime; pvs_init refuses two)."""
    import subprocess
    import sys
    pkg = os.path.join(REPO, "python-visual-similarity_amd")
    first = "from pvsim import _ffi; _ffi.lib()" if order == "engine_first" else "import torch"
    second = "import torch" if order == "engine_first" else "from pvsim import _ffi; _ffi.lib()"
    code = (f"import sys; sys.path.insert(0, {pkg!r}); {first}; {second}; from pvsim import _ffi; "
            "import torch; print('RUNTIMES', len(_ffi.mapped_hip_runtimes()), torch.cuda.is_available())")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RUNTIMES")][-1].split()
    assert line[1] == "1", r.stdout
