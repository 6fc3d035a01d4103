"""Pins the CPU oracle (oracle/pvsim_oracle.py) to golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import pvsim_oracle as orc
from pvsim import synth
from conftest import load_golden

# Tolerances (documented in DESIGN.md section "Parity"):
#  * labels: bit-exact (integer) except where the exact fp64 best/second gap is below fp32 noise
#  * VLAD values fp32: reference sums sequentially in fp32; oracle does the same -> <= 2e-7 abs
#    (residual difference is np.linalg.norm's summation order)
#  * Fisher fp64: <= 1e-12 abs   * cosine fp32: <= 3e-7 abs, fp64: <= 1e-14
VLAD_ATOL = 2e-7
FISHER_ATOL = 1e-12


def _images(g, key_raw="raw_u8", key_off="offsets"):
    return orc.split_ragged(g[key_raw].astype(np.float32), g[key_off])


def test_vlad_k256_labels_and_values(tables):
    g = load_golden("vlad_k256_d128")
    C = tables["centroids"]
    imgs = [synth.rootsift(r) for r in _images(g)]
    labels = np.concatenate([orc.kmeans_predict(x, C) for x in imgs])
    assert np.array_equal(labels, g["labels"])            # bit-exact index parity
    V = orc.vlad_encode(imgs, C)
    assert V.dtype == np.float32 and V.shape == g["vlad"].shape == (8, 256 * 128)
    np.testing.assert_allclose(V, g["vlad"], rtol=0, atol=VLAD_ATOL)


def test_vlad_structure_known_answers(tables):
    # notebook shape known-answers: K*D and K+2KD layouts
    g = load_golden("vlad_k256_d128")
    assert g["vlad"].shape[1] == 256 * 128
    assert load_golden("fisher_k256_d128")["fisher"].shape[1] == 256 + 2 * 256 * 128
    # per-cluster rows are unit L2 or all-zero (intra-normalisation, no global L2)
    rows = g["vlad"].reshape(8, 256, 128)
    nrm = np.linalg.norm(rows, axis=2)
    assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm == 0))


def test_vlad_with_pca(tables):
    g = load_golden("vlad_k256_d128")
    gp = load_golden("vlad_pca64_k256")
    imgs = [synth.rootsift(r) for r in _images(g)][: int(gp["n_images"])]
    V = orc.vlad_encode(imgs, tables["centroids_pca64"], pca=(tables["pca_components"], tables["pca_mean"]))
    # PCA GEMM summation order differs from sklearn's -> a near-tie label may flip; none does here
    np.testing.assert_allclose(V, gp["vlad"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("tag,kw", [
    ("default", {}), ("p05", {"power": 0.5}), ("l1", {"norm_order": 1}),
    ("p03_l1", {"power": 0.3, "norm_order": 1}), ("eps", {"eps": 1e-3})])
def test_vlad_small_variants(tag, kw):
    g = load_golden("small_k16_d8")
    imgs = [r / np.float32(16.0) for r in orc.split_ragged(g["raw"], g["offsets"])]
    labels = np.concatenate([orc.kmeans_predict(x, g["centroids"]) for x in imgs])
    assert np.array_equal(labels, g["labels"])
    assert not np.any(labels == 5)      # centroid 5 duplicates centroid 3: first index wins
    V = orc.vlad_encode(imgs, g["centroids"], **kw)
    np.testing.assert_allclose(V, g["vlad_" + tag], rtol=0, atol=VLAD_ATOL)


def test_vlad_empty_image_quirk_is_fenced():
    g = load_golden("small_k16_d8")
    # reference: a batch with an empty image returns ONE 1-D zero vector (vlad.py:92-93)
    assert g["empty_quirk"].shape == (16 * 8,) and not g["empty_quirk"].any()
    # the engine (and oracle) define a zero ROW for that image instead
    assert not orc.vlad_encode_one(np.zeros((0, 8), np.float32), g["centroids"]).any()


def test_gmm_posterior(tables):
    g = load_golden("vlad_k256_d128")
    f = load_golden("fisher_k256_d128")
    x = synth.rootsift(_images(g)[3])
    r = orc.gmm_predict_proba(x, tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    assert r.dtype == np.float64
    np.testing.assert_allclose(r, f["resp_img3"], rtol=0, atol=1e-12)


def test_fisher_k256(tables):
    g = load_golden("vlad_k256_d128")
    f = load_golden("fisher_k256_d128")
    imgs = [synth.rootsift(r) for r in _images(g)]
    F = orc.fisher_encode([imgs[i] for i in f["image_index"]],
                          tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    assert F.dtype == np.float64
    np.testing.assert_allclose(F, f["fisher"], rtol=0, atol=FISHER_ATOL)
    np.testing.assert_allclose(np.linalg.norm(F, axis=1), 1.0, atol=1e-8)   # global L2


def test_fisher_with_pca(tables):
    g = load_golden("vlad_k256_d128")
    f = load_golden("fisher_pca64_k64")
    imgs = [synth.rootsift(r) for r in _images(g)][: int(f["n_images"])]
    F = orc.fisher_encode(imgs, tables["gmmp_weights"], tables["gmmp_means"], tables["gmmp_covariances"],
                          pca=(tables["pca_components"], tables["pca_mean"]))
    # the fp32 PCA GEMM's summation order differs from sklearn's BLAS call (~1e-7 on the projected
    # descriptors); the sqrt power-norm amplifies that on near-zero elements -> 2e-6 abs
    np.testing.assert_allclose(F, f["fisher"], rtol=0, atol=2e-6)


def test_fisher_deep_like():
    f = load_golden("fisher_deep_k32_d96")
    F = orc.fisher_encode(list(f["desc"]), f["gmm_weights"], f["gmm_means"], f["gmm_covariances"])
    np.testing.assert_allclose(F, f["fisher"], rtol=0, atol=FISHER_ATOL)


@pytest.mark.parametrize("tag,kw", [("default", {}), ("p1", {"power": 1.0}),
                                    ("l1", {"norm_order": 1}), ("p03", {"power": 0.3})])
def test_fisher_small_variants(tag, kw):
    g = load_golden("small_k16_d8")
    imgs = [r / np.float32(16.0) for r in orc.split_ragged(g["raw"], g["offsets"])]
    F = orc.fisher_encode(imgs, g["gmm_weights"], g["gmm_means"], g["gmm_covariances"], **kw)
    np.testing.assert_allclose(F, g["fisher_" + tag], rtol=0, atol=FISHER_ATOL)


def test_cosine():
    g = load_golden("cosine")
    c32 = orc.cosine_similarity(g["a32"], g["b32"])
    assert c32.dtype == np.float32
    np.testing.assert_allclose(c32, g["cos32"], rtol=0, atol=3e-7)
    assert not c32[:, 2].any()                                   # zero row stays zero
    assert np.array_equal(c32[:, 7], c32[:, 4])                  # duplicate rows tie exactly
    c64 = orc.cosine_similarity(g["a32"].astype(np.float64) * 1.7, g["b32"].astype(np.float64))
    assert c64.dtype == np.float64
    np.testing.assert_allclose(c64, g["cos64"], rtol=0, atol=1e-14)
    cm = orc.cosine_similarity(g["a32"], g["b32"].astype(np.float64))
    assert cm.dtype == np.float64
    np.testing.assert_allclose(cm, g["cos_mixed"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(orc.cosine_similarity(g["a32"][0], g["b32"][1]), g["cos_1d"], atol=3e-7)
    with pytest.raises(ValueError):
        orc.cosine_similarity(np.ones((2, 1), np.float32), np.ones((2, 1), np.float32))


def test_cosine_on_vlad_and_similarity_score(tables):
    g = load_golden("vlad_k256_d128")
    c = load_golden("cosine")
    np.testing.assert_allclose(orc.cosine_similarity(g["vlad"], g["vlad"]), c["vlad_self"], atol=5e-7)
    imgs = [synth.rootsift(r) for r in _images(g)]
    V = orc.vlad_encode(imgs, tables["centroids"])
    s = np.float32(orc.cosine_similarity(V[:3], V[2:7]))          # _base_encoder.py:371-385
    np.testing.assert_allclose(s, c["similarity_score_3x5"], atol=5e-7)


def test_pipeline_hstack(tables):
    g = load_golden("vlad_k256_d128")
    p = load_golden("pipeline")
    imgs = [synth.rootsift(r) for r in _images(g)][1:4]
    V = orc.vlad_encode(imgs, tables["centroids"])
    F = orc.fisher_encode(imgs, tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    P = np.hstack([V, F])                                           # pipeline.py:47-66
    assert P.shape == p["encoded"].shape
    np.testing.assert_allclose(P, p["encoded"], atol=VLAD_ATOL)
    np.testing.assert_allclose(np.float32(orc.cosine_similarity(P[:2], P[1:3])), p["score"], atol=5e-7)


def test_retrieval_and_eval(tables):
    g = load_golden("eval_db64")
    C = tables["centroids"]
    db = [synth.rootsift(r) for r in _images(g, "db_raw_u8", "db_offsets")]
    qs = [synth.rootsift(r) for r in _images(g, "q_raw_u8", "q_offsets")]
    dbv, qv = orc.vlad_encode(db, C), orc.vlad_encode(qs, C)
    sims = orc.cosine_similarity(qv, dbv)
    np.testing.assert_allclose(sims, g["sims"], atol=5e-7)
    # the golden is tie-free where it matters: inside the top-8 window consecutive sorted scores
    # are further apart than fp32 GEMM-order noise
    srt = -np.sort(-g["sims"], axis=1)
    gaps = srt[:, :-1] - srt[:, 1:]
    assert np.min(gaps[:, :8]) > 2e-6
    idx, val = orc.topk(sims, 7)
    assert np.array_equal(idx, g["top7_index"])                   # bit-identical top-k lists
    np.testing.assert_allclose(val, g["top7_score"], atol=5e-7)
    clear = gaps.min(axis=1) > 2e-6                               # rows whose full ranking is tie-free
    assert clear.sum() >= 6
    assert np.array_equal(np.argsort(-sims, axis=1, kind="stable")[clear], g["full_argsort"][clear])
    for k, key in ((1, "acc_k1"), (5, "acc_k5")):
        assert orc.top_k_accuracy(qv, g["q_labels"], dbv, g["db_labels"], k) == float(g[key])
    for k, key in ((None, "map_all"), (5, "map_k5"), (10, "map_k10")):
        assert abs(orc.top_k_map(qv, g["q_labels"], dbv, g["db_labels"], k) - float(g[key])) < 1e-12


def test_near_tie_labels_are_only_near_ties(tables):
    """Where fp32 labels differ from exact fp64 labels the fp64 margin must be below fp32 noise."""
    rng = np.random.default_rng(77)
    x = synth.rootsift(synth.sift_like(50000, rng))
    C = tables["centroids"]
    lab32 = orc.kmeans_predict(x, C)
    lab64, gap = orc.assignment_margin(x, C)
    bad = lab32 != lab64
    assert bad.sum() <= 5 and np.all(gap[bad] < 2e-6)


# ----------------------------------------------------------------------------- the C restatement
def test_c_oracle_vlad_matches_golden(tables):
    import pvsim_oracle_c as orc_c
    g = load_golden("vlad_k256_d128")
    for threads in (1, 0):
        V, labels = orc_c.vlad_encode(g["raw_u8"], g["offsets"], tables["centroids"], threads=threads,
                                      return_labels=True)
        assert np.array_equal(labels, g["labels"])
        np.testing.assert_allclose(V, g["vlad"], rtol=0, atol=VLAD_ATOL)


def test_c_oracle_variants_and_retrieval(tables):
    import pvsim_oracle_c as orc_c
    s = load_golden("small_k16_d8")
    x = (s["raw"] / np.float32(16.0)).astype(np.float32)
    for tag, kw in (("default", {}), ("p05", {"power": 0.5}), ("l1", {"norm_order": 1.0}),
                    ("p03_l1", {"power": 0.3, "norm_order": 1.0}), ("eps", {"eps": 1e-3})):
        V = orc_c.vlad_encode(x, s["offsets"], s["centroids"], **kw)
        np.testing.assert_allclose(V, s["vlad_" + tag], rtol=0, atol=5e-7)
    g = load_golden("eval_db64")
    dbv = orc_c.vlad_encode(g["db_raw_u8"], g["db_offsets"], tables["centroids"])
    qv = orc_c.vlad_encode(g["q_raw_u8"], g["q_offsets"], tables["centroids"])
    idx, val = orc_c.retrieve(qv, dbv, 7)
    assert np.array_equal(idx, g["top7_index"])
    np.testing.assert_allclose(val, g["top7_score"], atol=5e-7)


def test_learn_restatement_matches_the_reference_fits():
    """learn(): the restated PCA / Lloyd / EM procedures against the tables the reference's own learn() fitted
    (tests/golden/make_golden_learn.py)."""
    g = load_golden("learn_k16_d32")
    x = g["x_u8"].astype(np.float32) / np.float32(16.0)
    for tag, kw in (("km3", dict(max_iter=3, tol=0.0)), ("km", {})):
        c, labels, inertia, n_iter = orc.kmeans_lloyd(x, g["c0"], **kw)
        assert n_iter == int(g[tag + "_n_iter"]) and np.array_equal(labels, g[tag + "_labels"])
        np.testing.assert_allclose(c, g[tag + "_centers"], rtol=0, atol=5e-4)      # fp32 member sums, order differs
        assert abs(inertia - float(g[tag + "_inertia"])) <= 2e-5 * inertia
    comp, mean, ev = orc.pca_fit(x, 16)
    assert np.array_equal(comp, g["pca_components"]) and np.array_equal(mean, g["pca_mean"])
    np.testing.assert_allclose(ev, g["pca_explained_variance"], rtol=1e-6)
    w, mu, cov, lower, n_iter, conv = orc.gmm_em(x, g["g_w0"], g["g_m0"], 1.0 / g["g_p0"], max_iter=5, tol=0.0)
    assert n_iter == 5 and not conv and abs(lower - float(g["g5_lower"])) < 1e-11
    for a, b in ((w, "g5_weights"), (mu, "g5_means"), (cov, "g5_cov")):
        np.testing.assert_allclose(a, g[b], rtol=1e-10, atol=1e-12)
    w, mu, cov, lower, n_iter, conv = orc.gmm_em(x, g["g_w0"], g["g_m0"], 1.0 / g["g_p0"])
    assert n_iter == int(g["g_n_iter"]) and conv == bool(g["g_converged"])
    for a, b in ((w, "g_weights"), (mu, "g_means"), (cov, "g_cov")):
        np.testing.assert_allclose(a, g[b], rtol=1e-10, atol=1e-12)


# ----------------------------------------------------------------------------------- round 2 fixtures
def _config3():
    from config3_inputs import config3_inputs
    return config3_inputs(12)


def test_oracle_fisher_at_the_config3_shape():
    """BASELINE configs[2]: Fisher, D = 512, K = 256, n = 196 -- the NumPy restatement against the reference's own output
    (image 0 whole, every 32nd element of 12 images, the 12 x 12 float64 cosine matrix)."""
    g = load_golden("fisher_k256_d512")
    w, mu, cov, imgs = _config3()
    assert abs(sum(float(d.astype(np.float64).sum()) for d in imgs) - float(g["desc_checksum"])) < 1e-6   # same seeded inputs
    f = orc.fisher_encode(imgs, w, mu, cov)
    assert f.dtype == np.float64 and f.shape == (12, 256 + 2 * 256 * 512)
    np.testing.assert_allclose(f[0], g["image0"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(f[:, ::32], g["every32"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(orc.cosine_similarity(f, f), g["cos_self"], rtol=0, atol=1e-12)


def test_oracle_eval_with_fisher_vectors_ranks_in_float64(tables):
    g, e = load_golden("eval_fisher_db64"), load_golden("eval_db64")
    db = [orc.rootsift(r.astype(np.float32)) for r in orc.split_ragged(e["db_raw_u8"], e["db_offsets"])]
    qs = [orc.rootsift(r.astype(np.float32)) for r in orc.split_ragged(e["q_raw_u8"], e["q_offsets"])]
    fdb = orc.fisher_encode(db, tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    fq = orc.fisher_encode(qs, tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    sims = orc.cosine_similarity(fq, fdb)
    assert sims.dtype == np.float64
    np.testing.assert_allclose(sims, g["sims"], rtol=0, atol=1e-12)
    idx, val = orc.topk(sims, 7)
    assert np.array_equal(idx, g["top7_index"])
    np.testing.assert_allclose(val, g["top7_score"], rtol=0, atol=1e-12)
    assert np.array_equal(orc.argsort_desc(sims), g["full_argsort"])
    assert orc.top_k_accuracy(fq, e["q_labels"], fdb, e["db_labels"], 1) == float(g["acc_k1"])
    assert orc.top_k_accuracy(fq, e["q_labels"], fdb, e["db_labels"], 5) == float(g["acc_k5"])
    assert abs(orc.top_k_map(fq, e["q_labels"], fdb, e["db_labels"], None) - float(g["map_all"])) < 1e-12
    assert abs(orc.top_k_map(fq, e["q_labels"], fdb, e["db_labels"], 5) - float(g["map_k5"])) < 1e-12


# ------------------------------------------------------------------------------------------------- the reference's shipped tables
def test_oracle_on_the_reference_shipped_tables():
    """The restatement against the REFERENCE's encoders run on its OWN pretrained vocabularies (tests/golden/shipped_tables.npz,
    made by make_golden_shipped.py from the arrays extract_reference_tables.py read out of the shipped files without unpickling
    them): the RootSIFT mixture with 487 covariance entries at the reg_covar floor (precision 1e6), the PCA variants
    (128 -> 64, 514 -> 257: FV length 131,840 as in examples/pipeline.ipynb), and VLAD on the means_-derived codebook."""
    from shipped_inputs import shipped_inputs, MODELS
    from pvsim import synth
    g = load_golden("shipped_tables")
    raws, deep = shipped_inputs()
    imgs = [synth.rootsift(r.astype(np.float32)) for r in raws]
    t = np.load(os.path.join(MODELS, "gmm_k256_root_sift_no_pca.npz"), allow_pickle=False)
    assert int((t["covariances"] < 1.01e-6).sum()) == 487
    F = orc.fisher_encode(imgs, t["weights"], t["means"], t["covariances"])
    np.testing.assert_allclose(F[0], g["fisher_rootsift_img0"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(F[:, ::16], g["fisher_rootsift_every16"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(orc.cosine_similarity(F, F), g["fisher_rootsift_cos"], rtol=0, atol=1e-12)
    tp = np.load(os.path.join(MODELS, "gmm_k256_root_sift_pca.npz"), allow_pickle=False)
    pp = np.load(os.path.join(MODELS, "pca_k256_root_sift_f2.npz"), allow_pickle=False)
    Fp = orc.fisher_encode([orc.pca_transform(x, pp["components"], pp["mean"]) for x in imgs], tp["weights"], tp["means"], tp["covariances"])
    np.testing.assert_allclose(Fp[:, ::16], g["fisher_rootsift_pca_every16"], rtol=0, atol=1e-7)     # fp32 projection: BLAS order
    td = np.load(os.path.join(MODELS, "gmm_k256_deep_features_vgg16_pca.npz"), allow_pickle=False)
    pd_ = np.load(os.path.join(MODELS, "pca_k256_deep_features_vgg16_f2.npz"), allow_pickle=False)
    Fd = orc.fisher_encode([orc.pca_transform(x, pd_["components"], pd_["mean"]) for x in deep], td["weights"], td["means"], td["covariances"])
    assert Fd.shape == (4, 131840)
    np.testing.assert_allclose(Fd[:, ::16], g["fisher_vgg16_pca_every16"], rtol=0, atol=2e-6)        # 514-term fp32 projections
    C = np.load(os.path.join(MODELS, "k_means_k256_root_sift_no_pca.npz"), allow_pickle=False)["cluster_centers"]
    lab = np.concatenate([orc.kmeans_predict(x, C) for x in imgs])
    assert np.array_equal(lab, g["vlad_labels"])
    np.testing.assert_allclose(orc.vlad_encode(imgs, C), g["vlad_rootsift"], rtol=0, atol=2e-7)


@pytest.mark.skipif(not os.path.isdir("/root/reference/pyvisim/res/model_files"), reason="the reference checkout is only present in the build container")
def test_shipped_table_files_equal_a_fresh_non_executing_read_of_the_reference():
    """tests/golden/extract_reference_tables.py --check: the committed pvsim/res/model_files/*.npz are what the opcode walk
    (no unpickling, no class lookups) reads out of the reference's own files today; a class record outside its allow-list
    or an opcode outside its subset would stop it."""
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "golden", "extract_reference_tables.py"), "--check"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("same") == 13 and "DIFFERS" not in r.stdout          # 8 shipped files + 5 derived codebooks


def test_the_table_reader_refuses_what_it_does_not_know(tmp_path):
    """The reader is not an unpickler: a stream that names any other class, or uses an opcode outside the subset, is an error."""
    import pickle
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    import extract_reference_tables as ext
    p = tmp_path / "evil.pkl"
    p.write_bytes(pickle.dumps(os.system, protocol=4))                         # a global the allow-list does not hold
    with pytest.raises(ValueError, match="unexpected class record"):
        ext.read_tables(str(p))
    p.write_bytes(pickle.dumps({"a": 1.5, "b": [1, 2]}, protocol=0))           # protocol-0 opcodes are outside the subset
    with pytest.raises(ValueError):
        ext.read_tables(str(p))
