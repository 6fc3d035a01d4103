"""GPU parity tests: the HIP path, called through the C-ABI (pvsim.Context is the ctypes binding of
include/pvsim.h), against the golden vectors of the reference and against the CPU oracle on seeded inputs.

Tolerances (also in DESIGN.md):
  labels / top-k indices   bit-exact (near-tie descriptors excepted, see test_labels_large_only_near_ties)
  VLAD values (fp32)       5e-7 abs   (sums are in reference order; only the norm reduction order differs)
  Fisher values (fp64)     1e-9 abs   (device computes in fp64; reduction order differs from NumPy's BLAS)
  cosine (fp32)            2e-6 abs   (fp32 GEMM summation order; L up to 65792)
"""
import numpy as np
import pytest

import pvsim_oracle as orc
from conftest import load_golden
from pvsim import synth, pack_descriptors, _ffi
from pvsim.engine import DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT

pytestmark = pytest.mark.gpu

VLAD_ATOL = 5e-7
FISHER_ATOL = 1e-9
COS_ATOL = 2e-6


def _raws(g, key_raw="raw_u8", key_off="offsets"):
    return orc.split_ragged(g[key_raw], g[key_off])


# ======================================================================================= VLAD
@pytest.mark.parametrize("kind", [DESC_U8_ROOTSIFT, DESC_F32_ROOTSIFT, DESC_F32])
def test_vlad_golden_k256(gpu_ctx, tables, kind):
    g = load_golden("vlad_k256_d128")
    cb = gpu_ctx.codebook(tables["centroids"])
    if kind == DESC_U8_ROOTSIFT:
        packed = g["raw_u8"]
    elif kind == DESC_F32_ROOTSIFT:
        packed = g["raw_u8"].astype(np.float32)
    else:
        packed = synth.rootsift(g["raw_u8"].astype(np.float32))
    v, labels = gpu_ctx.vlad_encode(cb, packed, g["offsets"], kind, return_labels=True)
    assert np.array_equal(labels, g["labels"])
    assert v.shape == g["vlad"].shape and v.dtype == np.float32
    np.testing.assert_allclose(v, g["vlad"], rtol=0, atol=VLAD_ATOL)


@pytest.mark.parametrize("tag,kw", [
    ("default", {}), ("p05", {"power": 0.5}), ("l1", {"norm_order": 1}),
    ("p03_l1", {"power": 0.3, "norm_order": 1}), ("eps", {"epsilon": 1e-3})])
def test_vlad_small_variants(gpu_ctx, tag, kw):
    g = load_golden("small_k16_d8")
    cb = gpu_ctx.codebook(g["centroids"])
    x = (g["raw"] / np.float32(16.0)).astype(np.float32)
    v, labels = gpu_ctx.vlad_encode(cb, x, g["offsets"], DESC_F32, return_labels=True, **kw)
    assert np.array_equal(labels, g["labels"])          # incl. the duplicated-centroid first-min tie
    np.testing.assert_allclose(v, g["vlad_" + tag], rtol=0, atol=2e-6 if "p03" in tag else VLAD_ATOL)


def test_vlad_pca_golden(gpu_ctx, tables):
    g = load_golden("vlad_k256_d128")
    gp = load_golden("vlad_pca64_k256")
    n = int(gp["n_images"])
    cb = gpu_ctx.codebook(tables["centroids_pca64"])
    pca = gpu_ctx.pca(tables["pca_components"], tables["pca_mean"])
    off = g["offsets"][: n + 1]
    v = gpu_ctx.vlad_encode(cb, g["raw_u8"][: off[-1]], off, DESC_U8_ROOTSIFT, pca=pca)
    np.testing.assert_allclose(v, gp["vlad"], rtol=0, atol=5e-6)     # fp32 PCA GEMM order (see oracle test)


@pytest.mark.parametrize("K,D", [(256, 128), (40, 100), (300, 64), (64, 30), (32, 200), (256, 514)])
def test_vlad_shapes_vs_oracle(gpu_ctx, K, D):
    """Ragged images incl. empty, single-descriptor and > 4096-descriptor (chunk continuation) ones; K / D
    that exercise cluster padding (K=40), two cluster blocks (K=300), scalar loads (D=30), two dim slabs
    (D=200, 514)."""
    rng = np.random.default_rng(K * 1000 + D)
    C = rng.normal(size=(K, D)).astype(np.float32)
    counts = [0, 1, 5, 33, 257, 1000, 4500 if D <= 128 else 300, 64]
    imgs = [(C[rng.integers(0, K, n)] + 0.7 * rng.normal(size=(n, D))).astype(np.float32) for n in counts]
    packed, offsets = pack_descriptors(imgs, D)
    cb = gpu_ctx.codebook(C)
    v, labels = gpu_ctx.vlad_encode(cb, packed, offsets, DESC_F32, return_labels=True)
    ref_labels = orc.kmeans_predict(packed, C)
    lab64, gap = orc.assignment_margin(packed, C)
    bad = labels != ref_labels
    # only near ties may differ: the fp64 gap of the two nearest centres must be below what a D-term fp32 evaluation of
    # |c|^2 - 2 x.c resolves (worst case (D + 4) 2^-23 (|x||c| + |c|^2) on either side)
    cmax = float(np.linalg.norm(C, axis=1).max())
    lim = 2.0 * (D + 4) * 2.0 ** -23 * (np.linalg.norm(packed[bad].astype(np.float64), axis=1) * cmax + cmax * cmax)
    assert np.all(gap[bad] < lim), (gap[bad] / lim).max()
    assert bad.mean() < 2e-3
    # values: every image whose labels all agree with the oracle's is compared with the oracle end to end (labels AND sums);
    # an image holding a flipped near tie is compared with the oracle's aggregate of the device labels
    n_full = 0
    for i, (x, o) in enumerate(zip(imgs, offsets[:-1])):
        if not len(x):
            assert not v[i].any()                                       # empty image -> zero row
            continue
        same = not bad[o:o + len(x)].any()
        lab = ref_labels[o:o + len(x)] if same else labels[o:o + len(x)]
        ref = orc.vlad_normalise(orc.vlad_aggregate(x, lab, C), 1, 2, 1e-9).reshape(-1)
        np.testing.assert_allclose(v[i], ref, rtol=0, atol=2e-6)
        n_full += same
    assert n_full >= len(imgs) - 3


def test_labels_large_only_near_ties(gpu_ctx, tables):
    """200k RootSIFT descriptors: labels equal the fp32 oracle's except where the exact fp64 margin
    between the two best centroids is below fp32 resolution (any fp32 evaluation may differ there)."""
    rng = np.random.default_rng(2024)
    raw = synth.sift_like(200_000, rng)
    x = synth.rootsift(raw)
    C = tables["centroids"]
    cb = gpu_ctx.codebook(C)
    _, labels = gpu_ctx.vlad_encode(cb, raw.astype(np.uint8), np.array([0, len(raw)], np.int64), DESC_U8_ROOTSIFT,
                                    return_labels=True)
    ref = orc.kmeans_predict(x, C)
    lab64, gap = orc.assignment_margin(x, C)
    bad = labels != ref
    print(f"label mismatches vs fp32 oracle: {bad.sum()} / {len(ref)}; vs exact fp64: {(labels != lab64).sum()}")
    assert bad.sum() <= 20 and np.all(gap[bad] < 5e-6)
    assert np.all(gap[labels != lab64] < 5e-6)
    # and with NO exception against the device's own defined recurrence restated in C (fma chain in the MFMA's dim order,
    # fmaf(-2, dot, |c|^2), strict '<'): near ties included
    import pvsim_oracle_c as orc_c
    assert np.array_equal(labels, orc_c.assign_chain(x, C))


def test_vlad_bitwise_reproducible(gpu_ctx, tables):
    rng = np.random.default_rng(5)
    raws = [synth.sift_like(int(n), rng).astype(np.uint8) for n in rng.integers(1, 900, 64)]
    packed, offsets = pack_descriptors(raws, 128, np.uint8)
    cb = gpu_ctx.codebook(tables["centroids"])
    a = gpu_ctx.vlad_encode(cb, packed, offsets, DESC_U8_ROOTSIFT)
    b = gpu_ctx.vlad_encode(cb, packed, offsets, DESC_U8_ROOTSIFT)
    assert np.array_equal(a, b)                                          # no float atomics anywhere
    rows = a.reshape(64, 256, 128)
    nrm = np.linalg.norm(rows, axis=2)
    assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm == 0))                 # intra-normalisation property


# ======================================================================================= Fisher
def test_fisher_golden_k256(gpu_ctx, tables):
    g = load_golden("vlad_k256_d128")
    f = load_golden("fisher_k256_d128")
    raws = _raws(g)
    sel = [raws[i] for i in f["image_index"]]
    gm = gpu_ctx.gmm(tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    packed, offsets = pack_descriptors(sel, 128, np.uint8)
    out = gpu_ctx.fisher_encode(gm, packed, offsets, DESC_U8_ROOTSIFT)
    assert out.dtype == np.float64 and out.shape == f["fisher"].shape
    np.testing.assert_allclose(out, f["fisher"], rtol=0, atol=FISHER_ATOL)


@pytest.mark.parametrize("tag,kw", [("default", {}), ("p1", {"power": 1.0}), ("l1", {"norm_order": 1}),
                                    ("p03", {"power": 0.3})])
def test_fisher_small_variants(gpu_ctx, tag, kw):
    g = load_golden("small_k16_d8")
    gm = gpu_ctx.gmm(g["gmm_weights"], g["gmm_means"], g["gmm_covariances"])
    x = (g["raw"] / np.float32(16.0)).astype(np.float32)
    out = gpu_ctx.fisher_encode(gm, x, g["offsets"], DESC_F32, **kw)
    np.testing.assert_allclose(out, g["fisher_" + tag], rtol=0, atol=FISHER_ATOL)


def test_fisher_deep_like_and_pca(gpu_ctx, tables):
    f = load_golden("fisher_deep_k32_d96")
    gm = gpu_ctx.gmm(f["gmm_weights"], f["gmm_means"], f["gmm_covariances"])
    packed, offsets = pack_descriptors(list(f["desc"]), 96)
    np.testing.assert_allclose(gpu_ctx.fisher_encode(gm, packed, offsets), f["fisher"], rtol=0, atol=FISHER_ATOL)
    g = load_golden("vlad_k256_d128")
    fp = load_golden("fisher_pca64_k64")
    n = int(fp["n_images"])
    gmp = gpu_ctx.gmm(tables["gmmp_weights"], tables["gmmp_means"], tables["gmmp_covariances"])
    pca = gpu_ctx.pca(tables["pca_components"], tables["pca_mean"])
    off = g["offsets"][: n + 1]
    out = gpu_ctx.fisher_encode(gmp, g["raw_u8"][: off[-1]], off, DESC_U8_ROOTSIFT, pca=pca)
    np.testing.assert_allclose(out, fp["fisher"], rtol=0, atol=2e-6)       # fp32 PCA GEMM order


# ======================================================================================= cosine
def test_cosine_golden(gpu_ctx):
    g = load_golden("cosine")
    c32 = gpu_ctx.cosine(g["a32"], g["b32"])
    assert c32.dtype == np.float32
    np.testing.assert_allclose(c32, g["cos32"], rtol=0, atol=COS_ATOL)
    assert not c32[:, 2].any() and np.array_equal(c32[:, 7], c32[:, 4])
    c64 = gpu_ctx.cosine(g["a32"].astype(np.float64) * 1.7, g["b32"].astype(np.float64))
    assert c64.dtype == np.float64
    np.testing.assert_allclose(c64, g["cos64"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(gpu_ctx.cosine(g["a32"], g["b32"].astype(np.float64)), g["cos_mixed"], atol=1e-12)
    v = load_golden("vlad_k256_d128")["vlad"]
    np.testing.assert_allclose(gpu_ctx.cosine(v, v), g["vlad_self"], rtol=0, atol=COS_ATOL)


@pytest.mark.parametrize("M,N,L", [(1, 1, 2), (3, 200, 40), (130, 257, 32768), (129, 128, 100), (64, 70, 37),
                                   (300, 5, 65792)])
def test_cosine_shapes_vs_oracle(gpu_ctx, M, N, L):
    rng = np.random.default_rng(M * 7 + N * 3 + L)
    a = rng.normal(size=(M, L)).astype(np.float32)
    b = rng.normal(size=(N, L)).astype(np.float32)
    if N > 2:
        b[1] = 0.0
    np.testing.assert_allclose(gpu_ctx.cosine(a, b), orc.cosine_similarity(a, b), rtol=0, atol=COS_ATOL)


# ======================================================================================= top-k
def _check_topk(gpu_ctx, q, db, k):
    idx, val = gpu_ctx.cosine_topk(q, db, k)
    s = gpu_ctx.cosine(q, db)                        # rank the device's own scores: isolates the select kernel
    ridx, rval = orc.topk(s, k)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(val, rval)
    # ... and against the oracle's own scores: the same index at every rank whose oracle score is separated from both
    # neighbours by more than the two evaluations can differ (2e-6 each), the same value within that tolerance everywhere
    so = orc.cosine_similarity(q, db).astype(np.float64)
    order = np.argsort(-so, axis=1, kind="stable")[:, :min(k + 1, so.shape[1])]
    sv = np.take_along_axis(so, order, axis=1)
    kk = min(k, so.shape[1])
    np.testing.assert_allclose(val[:, :kk], sv[:, :kk], rtol=0, atol=3e-6)
    gap_prev = np.hstack([np.full((len(sv), 1), np.inf), sv[:, :-1] - sv[:, 1:]])[:, :kk]
    gap_next = np.hstack([sv[:, :-1] - sv[:, 1:], np.full((len(sv), 1), np.inf)])[:, :kk]
    clear = (gap_prev > 5e-6) & (gap_next > 5e-6)
    assert np.array_equal(idx[:, :kk][clear], order[:, :kk][clear])
    assert clear.mean() > 0.5


@pytest.mark.parametrize("nq,N,k", [(5, 10, 3), (7, 8189, 5), (3, 8189, 100), (2, 20000, 1000), (4, 300, 300),
                                    (2, 40000, 17)])
def test_topk_vs_stable_argsort(gpu_ctx, nq, N, k):
    rng = np.random.default_rng(nq * N + k)
    L = 64
    db = rng.normal(size=(N, L)).astype(np.float32)
    db[N // 2] = db[N // 3]                          # a duplicate row -> exact score tie, index order decides
    db[N - 1] = 0.0                                  # a zero row
    q = rng.normal(size=(nq, L)).astype(np.float32)
    _check_topk(gpu_ctx, q, db, k)


def test_deep_ranking_pages(gpu_ctx):
    """k > 1024 (top_k_map(k=None) is a full argsort): paged select, equal to a stable argsort of the scores."""
    rng = np.random.default_rng(8)
    db = rng.normal(size=(5000, 32)).astype(np.float32)
    db[100] = db[7]; db[4000] = db[7]                  # ties across page boundaries are broken by index
    q = rng.normal(size=(3, 32)).astype(np.float32)
    for k in (1025, 2500, 5000):
        _check_topk(gpu_ctx, q, db, k)


def test_topk_all_ties_and_self(gpu_ctx):
    db = np.ones((500, 16), np.float32)              # every score identical: pure index order
    idx, _ = gpu_ctx.cosine_topk(db[:3], db, 9)
    assert np.array_equal(idx, np.tile(np.arange(9), (3, 1)))
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1000, 256)).astype(np.float32)
    idx, val = gpu_ctx.cosine_topk(x, x, 1)          # self retrieval: top-1 is the row itself
    assert np.array_equal(idx[:, 0], np.arange(1000)) and np.all(np.abs(val - 1) < 1e-5)


def test_retrieval_golden_topk_lists(gpu_ctx, tables):
    g = load_golden("eval_db64")
    cb = gpu_ctx.codebook(tables["centroids"])
    dbv = gpu_ctx.vlad_encode(cb, g["db_raw_u8"], g["db_offsets"], DESC_U8_ROOTSIFT)
    qv = gpu_ctx.vlad_encode(cb, g["q_raw_u8"], g["q_offsets"], DESC_U8_ROOTSIFT)
    np.testing.assert_allclose(gpu_ctx.cosine(qv, dbv), g["sims"], rtol=0, atol=COS_ATOL)
    idx, val = gpu_ctx.cosine_topk(qv, dbv, 7)
    assert np.array_equal(idx, g["top7_index"])       # bit-identical to the reference's lists
    np.testing.assert_allclose(val, g["top7_score"], rtol=0, atol=COS_ATOL)


# ======================================================================================= drop-in API
def _encoders(tables):
    from pvsim.encoders import VLADEncoder, FisherVectorEncoder
    from pvsim.features import Lambda
    from pvsim.models import KMeansModel, GMMModel
    fx = Lambda(synth.rootsift, 128)
    v = VLADEncoder(fx, kmeans_model=KMeansModel(tables["centroids"]))
    f = FisherVectorEncoder(fx, gmm_model=GMMModel(tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"]))
    return v, f


def test_api_encode_similarity_pipeline(tables):
    from pvsim.encoders import Pipeline
    g = load_golden("vlad_k256_d128")
    raws = [r.astype(np.float32) for r in _raws(g)]
    venc, fenc = _encoders(tables)
    np.testing.assert_allclose(venc.encode(raws), g["vlad"], rtol=0, atol=VLAD_ATOL)
    c = load_golden("cosine")
    s = venc.similarity_score(raws[:3], raws[2:7])
    assert s.dtype == np.float32 and s.shape == (3, 5)
    np.testing.assert_allclose(s, c["similarity_score_3x5"], rtol=0, atol=COS_ATOL)
    p = load_golden("pipeline")
    pipe = Pipeline([venc, fenc])
    enc = pipe.encode(raws[1:4])
    assert enc.shape == p["encoded"].shape
    np.testing.assert_allclose(enc, p["encoded"], rtol=0, atol=VLAD_ATOL)
    np.testing.assert_allclose(pipe.similarity_score(raws[1:3], raws[2:4]), p["score"], rtol=0, atol=COS_ATOL)
    # descriptor-level entry with fused RootSIFT from uint8 gives the same encodings
    np.testing.assert_allclose(venc.encode_descriptors(_raws(g), rootsift=True), g["vlad"], rtol=0, atol=VLAD_ATOL)
    # flatten=False stacks (K, D) blocks (vlad.py:110-115)
    venc.flatten = False
    assert venc.encode(raws[:2]).shape == (2 * 256, 128)


def test_api_eval_functions(tables):
    from pvsim import eval as ev
    g = load_golden("eval_db64")
    venc, _ = _encoders(tables)
    db = [r.astype(np.float32) for r in _raws(g, "db_raw_u8", "db_offsets")]
    qs = [r.astype(np.float32) for r in _raws(g, "q_raw_u8", "q_offsets")]
    paths = [f"img_{i:03d}.jpg" for i in range(len(db))]
    emap = dict(zip(paths, venc.encode(db)))
    plab = dict(zip(paths, [int(l) for l in g["db_labels"]]))
    top = ev.retrieve_top_k_similar([qs[0]], emap, venc, k=7)
    assert [paths.index(p) for p, _ in top] == list(g["top7_index"][0])
    np.testing.assert_allclose([s for _, s in top], g["top7_score"][0], atol=COS_ATOL)
    wrapped = [[q] for q in qs]
    labels = list(g["q_labels"])
    assert ev.top_k_accuracy(wrapped, labels, emap, plab, venc, 1) == float(g["acc_k1"])
    assert ev.top_k_accuracy(wrapped, labels, emap, plab, venc, 5) == float(g["acc_k5"])
    assert abs(ev.top_k_map(wrapped, labels, emap, plab, venc, 5) - float(g["map_k5"])) < 1e-12
    assert abs(ev.top_k_map(wrapped, labels, emap, plab, venc, 10) - float(g["map_k10"])) < 1e-12
    assert abs(ev.top_k_map(wrapped, labels, emap, plab, venc, None) - float(g["map_all"])) < 1e-12


# ======================================================================================= full-size properties
def test_config2_sized_properties(gpu_ctx, tables):
    """2048 ragged images (config-2 generator): oracle parity on a subsample, size-independent properties on
    all of it (unit / zero cluster rows, self-retrieval, symmetric scores, idempotent encode)."""
    rng = np.random.default_rng(1235)
    counts = synth.ragged_counts(2048, 1235)
    raws = [synth.sift_like(int(n), rng).astype(np.uint8) for n in counts]
    packed, offsets = pack_descriptors(raws, 128, np.uint8)
    cb = gpu_ctx.codebook(tables["centroids"])
    v = gpu_ctx.vlad_encode(cb, packed, offsets, DESC_U8_ROOTSIFT)
    sub = rng.choice(2048, 24, replace=False)
    ref = orc.vlad_encode([synth.rootsift(raws[i]) for i in sub], tables["centroids"])
    np.testing.assert_allclose(v[sub], ref, rtol=0, atol=VLAD_ATOL)
    nrm = np.linalg.norm(v.reshape(2048, 256, 128), axis=2)
    assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm == 0))
    idx, val = gpu_ctx.cosine_topk(v, v, 5)
    assert np.array_equal(idx[:, 0], np.arange(2048)) and np.all(np.abs(val[:, 0] - 1) < 1e-5)
    s = gpu_ctx.cosine(v[:300], v[:300])
    assert np.abs(s - s.T).max() < 1e-6
    ridx, _ = orc.topk(orc.cosine_similarity(v[sub], v), 5)
    assert np.array_equal(idx[sub], ridx)


# ======================================================================================= multi-GPU decomposition
@pytest.mark.parametrize("world,n_total", [(3, 1000), (8, 8189 // 4), (2, 257)])
def test_sharded_decomposition_on_one_gpu(gpu_ctx, world, n_total):
    """The N-rank decomposition of bench.py / pvsim.distributed run rank by rank on ONE GPU (the all-gather is
    emulated by laying the padded blocks out as all_gather_into_tensor would): uneven last block, true global
    indices, running top-k merge across blocks -- must equal the single-GPU answer exactly."""
    import torch
    from pvsim import distributed as pd
    rng = np.random.default_rng(world * 100 + n_total)
    L, k = 512, 6
    enc = rng.normal(size=(n_total, L)).astype(np.float32)
    enc[n_total // 2] = enc[3]                                   # a tie that straddles blocks
    ref_idx, ref_val = gpu_ctx.cosine_topk(enc, enc, k)
    dev = torch.device("cuda", 0)
    _, _, block = pd.shard_range(n_total, world, 0)
    enc_all = torch.zeros((world * block, L), dtype=torch.float32, device=dev)
    inv_all = torch.ones((world * block,), dtype=torch.float32, device=dev)
    for r in range(world):
        lo, hi, _ = pd.shard_range(n_total, world, r)
        enc_all[r * block: r * block + hi - lo] = torch.from_numpy(enc[lo:hi]).to(dev)
    torch.cuda.synchronize()
    gpu_ctx.row_inv_norms_dev(enc_all.data_ptr(), world * block, L, inv_all.data_ptr())
    gpu_ctx.sync()
    for r in range(world):                                       # what each rank does before the gather
        lo, hi, _ = pd.shard_range(n_total, world, r)
        pd.mask_padding(inv_all[r * block:(r + 1) * block], hi - lo)
    torch.cuda.synchronize()
    score = pd.device_score_block(gpu_ctx)
    got_idx, got_val = [], []
    for r in range(world):
        lo, hi, _ = pd.shard_range(n_total, world, r)
        idx = torch.full((block, k), -1, dtype=torch.int64, device=dev)
        val = torch.full((block, k), float("-inf"), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        n_loc = pd.retrieve_sharded(enc_all[r * block:], inv_all[r * block:], enc_all, inv_all, n_total, r, world, k,
                                    score, idx, val)
        gpu_ctx.sync()
        assert n_loc == hi - lo
        got_idx.append(idx[:n_loc].cpu().numpy())
        got_val.append(val[:n_loc].cpu().numpy())
    assert np.array_equal(np.concatenate(got_idx), ref_idx)
    assert np.array_equal(np.concatenate(got_val), ref_val)


# ======================================================================================= fp16 similarity (config 5)
def test_fp16_cosine_and_recall(gpu_ctx, tables):
    """fp16 operands, fp32 accumulate.  (1) exact check: on fp16-rounded inputs the kernel equals the fp64 dot of
    those rounded values to fp32-accumulation accuracy; (2) recall@10 of the fp16 ranking against the exact fp32
    ranking on VLAD encodings (BASELINE configs[4] correctness criterion)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(55)
    counts = rng.integers(100, 700, 700)
    raws = [synth.sift_like(int(n), rng).astype(np.uint8) for n in counts]
    packed, offsets = pack_descriptors(raws, 128, np.uint8)
    cb = gpu_ctx.codebook(tables["centroids"])
    v = gpu_ctx.vlad_encode(cb, packed, offsets, DESC_U8_ROOTSIFT)           # (700, 32768) fp32
    n, L = v.shape
    t32 = torch.from_numpy(v).to(dev)
    t16 = torch.empty((n, L), dtype=torch.float16, device=dev)
    inv = torch.empty((n,), dtype=torch.float32, device=dev)
    out = torch.empty((n, n), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.f32_to_f16_dev(t32.data_ptr(), n * L, t16.data_ptr())
    gpu_ctx.row_inv_norms_dev(t32.data_ptr(), n, L, inv.data_ptr())
    gpu_ctx.cosine_f16_dev(t16.data_ptr(), n, t16.data_ptr(), n, L, inv.data_ptr(), inv.data_ptr(), out.data_ptr(), n)
    gpu_ctx.sync()
    assert np.array_equal(t16.cpu().numpy(), v.astype(np.float16))          # conversion is round-to-nearest-even
    h = v.astype(np.float16).astype(np.float64)
    iv = inv.cpu().numpy().astype(np.float64)
    ref16 = (h @ h.T) * iv[:, None] * iv[None, :]
    got = out.cpu().numpy()
    assert np.abs(got - ref16).max() < 5e-6
    assert np.array_equal(got, got.T)                                       # symmetric mode: bitwise symmetric
    # general (non-symmetric) launch agrees with the symmetric one
    out2 = torch.empty((100, n), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.cosine_f16_dev(t16[300:].data_ptr(), 100, t16.data_ptr(), n, L, inv[300:].data_ptr(), inv.data_ptr(),
                           out2.data_ptr(), n)
    gpu_ctx.sync()
    assert np.abs(out2.cpu().numpy() - got[300:400]).max() < 1e-6
    # recall@10 vs exact fp32
    k = 10
    idx16 = torch.empty((n, k), dtype=torch.int64, device=dev)
    val16 = torch.empty((n, k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.cosine_topk_f16_dev(t16.data_ptr(), n, t16.data_ptr(), n, L, inv.data_ptr(), inv.data_ptr(), k, 0, False,
                                idx16.data_ptr(), val16.data_ptr())
    gpu_ctx.sync()
    idx32, _ = gpu_ctx.cosine_topk(v, v, k)
    i16 = idx16.cpu().numpy()
    recall = np.mean([len(set(a) & set(b)) / k for a, b in zip(i16, idx32)])
    print(f"fp16 recall@{k} vs fp32: {recall:.4f}")
    assert recall >= 0.99 and np.array_equal(i16[:, 0], np.arange(n))


@pytest.mark.parametrize("world,n_total", [(2, 700), (3, 1000), (4, 1030), (8, 2047), (8, 1800), (5, 513)])
def test_symmetric_pair_scheme_on_one_gpu(gpu_ctx, world, n_total):
    """pvsim.distributed.retrieve_symmetric, all ranks run one after the other on ONE GPU with the all-to-all emulated
    by routing the send slabs: every block pair scored once (dual-store GEMM), lists exchanged and merged.  Must be
    BIT-identical (indices and values) to the single-GPU ranking, including uneven / empty trailing blocks."""
    import torch
    from pvsim import distributed as pd
    rng = np.random.default_rng(world * 1000 + n_total)
    L, k = 256, 7
    enc = rng.normal(size=(n_total, L)).astype(np.float32)
    enc[n_total // 2] = enc[5]                                   # a tie between different blocks
    ref_idx, ref_val = gpu_ctx.cosine_topk(enc, enc, k)
    dev = torch.device("cuda", 0)
    _, _, B = pd.shard_range(n_total, world, 0)
    enc_all = torch.zeros((world * B, L), dtype=torch.float32, device=dev)
    inv_all = torch.ones((world * B,), dtype=torch.float32, device=dev)
    for r in range(world):
        lo, hi, _ = pd.shard_range(n_total, world, r)
        enc_all[r * B: r * B + hi - lo] = torch.from_numpy(enc[lo:hi]).to(dev)
    torch.cuda.synchronize()
    gpu_ctx.row_inv_norms_dev(enc_all.data_ptr(), world * B, L, inv_all.data_ptr())
    gpu_ctx.sync()
    ops = pd.DeviceOps(gpu_ctx)

    def new_tensor(shape, dtype, fill):
        t_ = torch.full(shape, fill, dtype=getattr(torch, dtype), device=dev)
        torch.cuda.synchronize()
        return t_

    states = [pd.symmetric_local(enc_all, inv_all, n_total, r, world, k, ops, new_tensor) for r in range(world)]
    for r in range(world):                                       # all_to_all: slab p of rank r <- slab r of rank p
        for p in range(world):
            states[r]["m_idx"][p].copy_(states[p]["s_idx"][r])
            states[r]["m_val"][p].copy_(states[p]["s_val"][r])
    torch.cuda.synchronize()
    got_i, got_v = [], []
    for r in range(world):
        i_, v_ = pd.symmetric_finish(states[r], ops, new_tensor)
        got_i.append(i_.cpu().numpy()); got_v.append(v_.cpu().numpy())
    assert np.array_equal(np.concatenate(got_i), ref_idx)
    assert np.array_equal(np.concatenate(got_v), ref_val)

    # the same scheme without torch: pvs_malloc memory behind DevArray views, pool-allocated lists (pvsim.distributed)
    if world <= 3:
        pool = pd.DevicePool(gpu_ctx)
        d_enc = pool.empty((world * B, L), "float32").upload(enc_all.cpu().numpy())
        d_inv = pool.empty((world * B,), "float32").upload(inv_all.cpu().numpy())
        ops_s = pd.DeviceOps(gpu_ctx, same_stream=True)
        st2 = [pd.symmetric_local(d_enc, d_inv, n_total, r, world, k, ops_s, pool.full) for r in range(world)]
        gpu_ctx.sync()
        for r in range(world):
            for p_ in range(world):
                st2[r]["m_idx"][p_].upload(st2[p_]["s_idx"][r].numpy())
                st2[r]["m_val"][p_].upload(st2[p_]["s_val"][r].numpy())
        gi, gv = [], []
        for r in range(world):
            i_, v_ = pd.symmetric_finish(st2[r], ops_s, pool.full)
            gpu_ctx.sync()
            gi.append(i_.numpy()); gv.append(v_.numpy())
        assert np.array_equal(np.concatenate(gi), ref_idx) and np.array_equal(np.concatenate(gv), ref_val)
        pool.close()


# ======================================================================================= deep features (config 3 front end)
def test_deep_conv_feature_extractor_and_fisher_end_to_end():
    """DeepConvFeature on PyTorch-ROCm (own VGG16 `features` stack, random weights: no checkpoint offline): hook on the
    last Conv2d (pre-ReLU), 224x224 -> 14x14 = 196 descriptors of 512 (+2 spatial) dims, batch == per-image; then the
    drop-in FisherVectorEncoder on those descriptors equals the CPU restatement of the reference."""
    import torch
    from pvsim.features import DeepConvFeature
    from pvsim.encoders import FisherVectorEncoder
    from pvsim.models import GMMModel
    torch.manual_seed(0)
    fx = DeepConvFeature(spatial_encoding=True, device="cuda")
    assert fx.output_dim == 514 and fx.selected_layer_name.endswith("28")
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, size=(96 + 16 * i, 120, 3), dtype=np.uint8) for i in range(3)]
    d0 = fx(imgs[0])
    assert d0.shape == (196, 514) and d0.dtype == np.float32
    np.testing.assert_allclose(d0[:, 512], np.tile(np.arange(14) / 14, 14), atol=1e-7)      # x / W
    np.testing.assert_allclose(d0[:, 513], np.repeat(np.arange(14) / 14, 14), atol=1e-7)    # y / H
    batch = fx.batch(imgs).cpu().numpy()
    np.testing.assert_allclose(batch[0], d0, atol=1e-4)          # conv algorithms may differ between batch sizes
    K, D = 8, 514
    gm = GMMModel(np.full(K, 1.0 / K), rng.normal(0, 0.05, (K, D)), rng.uniform(0.01, 0.05, (K, D)))
    enc = FisherVectorEncoder(fx, gmm_model=gm)
    f = enc.encode(imgs)
    assert f.shape == (3, K + 2 * K * D) and f.dtype == np.float64
    ref = orc.fisher_encode([fx(im) for im in imgs], gm.weights_, gm.means_, gm.covariances_)
    np.testing.assert_allclose(f, ref, rtol=0, atol=1e-7)        # extractor output re-computed per call (conv noise)


# ======================================================================================= learn() on the device
# Tolerances: the reference's fits (scikit-learn) accumulate k-means sums in fp32 per thread chunk and PCA's covariance
# in fp32; the device forms them in fp64.  Labels must agree exactly; centres to 5e-4 abs (values up to 16, i.e. ~3e-5
# relative); GMM tables (fp64 on both sides) to 1e-9 relative; PCA axes to 2e-3 (the golden itself is fp32 noisy).
KM_ATOL = 5e-4


def _learn_rows(ctx, g):
    from pvsim import learn
    return learn.DeviceRows.from_host(ctx, g["x_u8"].astype(np.float32) / np.float32(16.0))


def test_learn_kmeans_matches_reference_fit(gpu_ctx):
    from pvsim import learn
    g = load_golden("learn_k16_d32")
    rows = _learn_rows(gpu_ctx, g)
    m = learn.fit_kmeans(rows, 16, init=g["c0"], n_init=1, max_iter=3, tol=0.0)
    assert m.n_iter_ == int(g["km3_n_iter"]) and np.array_equal(m.labels_, g["km3_labels"])
    np.testing.assert_allclose(m.cluster_centers_, g["km3_centers"], rtol=0, atol=KM_ATOL)
    assert abs(m.inertia_ - float(g["km3_inertia"])) <= 2e-5 * float(g["km3_inertia"])
    m = learn.fit_kmeans(rows, 16, init=g["c0"], n_init=1)               # default stopping rule
    assert m.n_iter_ == int(g["km_n_iter"]) and np.array_equal(m.labels_, g["km_labels"])
    np.testing.assert_allclose(m.cluster_centers_, g["km_centers"], rtol=0, atol=KM_ATOL)
    # and against the restatement run here on the same start
    c, l, inertia, n_it = orc.kmeans_lloyd(g["x_u8"].astype(np.float32) / 16, g["c0"])
    assert n_it == m.n_iter_ and np.array_equal(l, m.labels_)
    # run-to-run identical
    m2 = learn.fit_kmeans(rows, 16, init=g["c0"], n_init=1)
    assert np.array_equal(m2.cluster_centers_, m.cluster_centers_) and m2.inertia_ == m.inertia_
    rows.free()


def test_learn_gmm_matches_reference_fit(gpu_ctx):
    from pvsim import learn
    g = load_golden("learn_k16_d32")
    rows = _learn_rows(gpu_ctx, g)
    kw = dict(weights_init=g["g_w0"], means_init=g["g_m0"], precisions_init=g["g_p0"])
    with pytest.warns(UserWarning):                                        # tol=0 never converges, as in scikit-learn
        m = learn.fit_gmm(rows, 8, max_iter=5, tol=0.0, **kw)
    assert m.n_iter_ == 5 and not m.converged_
    np.testing.assert_allclose(m.weights_, g["g5_weights"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(m.means_, g["g5_means"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m.covariances_, g["g5_cov"], rtol=1e-9, atol=1e-11)
    assert abs(m.lower_bound_ - float(g["g5_lower"])) < 1e-10
    m = learn.fit_gmm(rows, 8, **kw)
    assert m.converged_ == bool(g["g_converged"]) and m.n_iter_ == int(g["g_n_iter"])
    np.testing.assert_allclose(m.means_, g["g_means"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m.covariances_, g["g_cov"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m.weights_, g["g_weights"], rtol=1e-9, atol=1e-13)
    rows.free()


def test_learn_pca_matches_reference_fit(gpu_ctx):
    from pvsim import learn
    g = load_golden("learn_k16_d32")
    rows = _learn_rows(gpu_ctx, g)
    m = learn.fit_pca(rows, 16)
    x = g["x_u8"].astype(np.float64) / 16
    np.testing.assert_allclose(m.mean_, g["pca_mean"], rtol=0, atol=2e-5)
    c64, mean64, ev64 = orc.pca_fit(x, 16)                                 # the same procedure in fp64: tight
    np.testing.assert_allclose(np.abs(np.sum(m.components_.astype(np.float64) * c64, axis=1)), 1.0, atol=1e-6)
    np.testing.assert_allclose(m.explained_variance_, ev64, rtol=1e-9)
    cos = np.sum(m.components_.astype(np.float64) * g["pca_components"].astype(np.float64), axis=1)
    assert np.all(cos > 1 - 2e-3), cos                                     # same axes AND same signs as the reference's fit
    np.testing.assert_allclose(m.explained_variance_, g["pca_explained_variance"], rtol=2e-3)
    # transform on the device with the fitted table
    red = rows.transformed(m)
    got = red.buf.download((rows.n, 16), np.float32)
    np.testing.assert_allclose(got, orc.pca_transform(x.astype(np.float32), m.components_, m.mean_), rtol=0, atol=2e-4)
    red.free()
    rows.free()


def test_learn_seeding_and_label_sums(gpu_ctx):
    from pvsim import learn
    g = load_golden("learn_k16_d32")
    x = g["x_u8"].astype(np.float32) / np.float32(16.0)
    rows = _learn_rows(gpu_ctx, g)
    c1, i1 = learn.kmeans_plusplus(rows, 16, random_state=3)
    c2, i2 = learn.kmeans_plusplus(rows, 16, random_state=3)
    assert np.array_equal(i1, i2) and np.array_equal(c1, c2) and len(set(i1.tolist())) == 16
    # the run without host round trips (one call) and the stepwise entry points: the same random stream -> the same centres
    for K_, rs in ((16, 3), (5, 11), (40, 7)):
        _, ia = learn.kmeans_plusplus(rows, K_, random_state=rs)
        _, ib = learn.kmeans_plusplus(rows, K_, random_state=rs, stepwise=True)
        assert np.array_equal(ia, ib), (K_, rs, ia, ib)
    assert np.array_equal(c1, x[i1])
    # seeded k-means++ start + Lloyd reaches an inertia close to the reference's fit from its own start
    m = learn.fit_kmeans(rows, 16, random_state=3)
    assert m.inertia_ < 1.05 * float(g["km_inertia"])
    # potentials: sum_i min(mind, d_j) against NumPy
    cand = x[[5, 77, 4000]]
    dist = gpu_ctx.buffer(3 * rows.n * 4)
    pot = gpu_ctx.seed_distances_dev(rows.ptr, 32, rows.n, cand, None, dist.ptr)
    d = ((x[None, :, :].astype(np.float64) - cand[:, None, :]) ** 2).sum(-1)
    np.testing.assert_allclose(pot, d.sum(1), rtol=1e-6)
    np.testing.assert_allclose(dist.download((3, rows.n), np.float32), d, rtol=2e-6, atol=1e-6)
    # per-label sums
    lab = gpu_ctx.buffer(rows.n * 4).upload(m.labels_)
    s1 = gpu_ctx.label_sums_dev(rows.ptr, 32, rows.n, lab.ptr, 16, square=False)
    s2 = gpu_ctx.label_sums_dev(rows.ptr, 32, rows.n, lab.ptr, 16, square=True)
    r1, r2 = np.zeros((16, 32)), np.zeros((16, 32))
    np.add.at(r1, m.labels_, x.astype(np.float64))
    np.add.at(r2, m.labels_, (x * x).astype(np.float64))
    np.testing.assert_allclose(s1, r1, rtol=5e-6)       # fp32 sums inside a 4096-descriptor chunk, fp64 across chunks
    np.testing.assert_allclose(s2, r2, rtol=5e-6)
    # GMM from the k-means start (scikit-learn's default init_params): a valid mixture at least as good as the golden's
    gm = learn.fit_gmm(rows, 8, random_state=0)
    assert abs(gm.weights_.sum() - 1) < 1e-12 and np.all(gm.covariances_ > 0) and gm.converged_
    assert gm.lower_bound_ > float(g["g_lower"]) - 1.0
    for b in (dist, lab):
        b.free()
    rows.free()


def test_learn_through_the_encoder_api(tables):
    """VLADEncoder.learn / FisherVectorEncoder.learn as the reference calls them (keyword arguments of the scikit-learn
    estimators), including dim_reduction_factor, then encode with the learnt vocabulary."""
    from pvsim.encoders import VLADEncoder, FisherVectorEncoder
    from pvsim.features import Lambda
    from pvsim.models import KMeansModel, GMMModel
    g = load_golden("learn_k16_d32")
    xu8 = g["x_u8"]
    images = [xu8[i * 500:(i + 1) * 500].astype(np.int64) for i in range(40)]
    fx = Lambda(lambda im: im.astype(np.float32) / np.float32(16.0), 32)
    enc = VLADEncoder(fx, kmeans_model=KMeansModel(g["c0"]))
    enc.learn(images, n_clusters=16, init=g["c0"], n_init=1, max_iter=3, tol=0.0)
    np.testing.assert_allclose(enc.clustering_model.cluster_centers_, g["km3_centers"], rtol=0, atol=KM_ATOL)
    v = enc.encode(images[:3])
    want = orc.vlad_encode([fx(im) for im in images[:3]], enc.clustering_model.cluster_centers_)
    np.testing.assert_allclose(v, want, rtol=0, atol=VLAD_ATOL)
    enc.learn(images, n_clusters=16, dim_reduction_factor=2, n_init=1, random_state=0, max_iter=5)
    assert enc.pca is not None and enc.pca.n_components == 16 and enc.clustering_model.n_features_in_ == 16
    assert enc.encode(images[:2]).shape == (2, 16 * 16)
    with pytest.raises(TypeError):
        enc.learn(images, n_clusters=16, not_a_kmeans_argument=1)
    fe = FisherVectorEncoder(fx, gmm_model=GMMModel(np.full(8, 1 / 8), g["g_m0"], 1 / g["g_p0"]))
    fe.learn(images, n_clusters=8, weights_init=g["g_w0"], means_init=g["g_m0"], precisions_init=g["g_p0"])
    np.testing.assert_allclose(fe.clustering_model.means_, g["g_means"], rtol=1e-9, atol=1e-11)
    f = fe.encode(images[:2])
    want = orc.fisher_encode([fx(im) for im in images[:2]], fe.clustering_model.weights_, fe.clustering_model.means_,
                             fe.clustering_model.covariances_)
    np.testing.assert_allclose(f, want, rtol=0, atol=FISHER_ATOL)


# ======================================================================================= filtered (prefilter + exact re-score) top-k
def _topk_both(ctx, q, db, k, same):
    import torch
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy(q).to(dev)
    tdb = tq if same else torch.from_numpy(db).to(dev)
    nq, L = tq.shape
    N = tdb.shape[0]
    iq = torch.empty((nq,), dtype=torch.float32, device=dev)
    idb = iq if same else torch.empty((N,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.row_inv_norms_dev(tq.data_ptr(), nq, L, iq.data_ptr())
    if not same:
        ctx.row_inv_norms_dev(tdb.data_ptr(), N, L, idb.data_ptr())
    out = []
    for filt in (False, True):
        idx = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
        val = torch.full((nq, k), -7.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        if filt:
            st = ctx.cosine_topk_filtered_dev(tq.data_ptr(), nq, tdb.data_ptr(), N, L, iq.data_ptr(), idb.data_ptr(), k,
                                              idx.data_ptr(), val.data_ptr())
        else:
            st = None
            ctx.cosine_topk_dev(tq.data_ptr(), nq, tdb.data_ptr(), N, L, iq.data_ptr(), idb.data_ptr(), k, 0, False,
                                idx.data_ptr(), val.data_ptr())
        ctx.sync()
        out.append((idx.cpu().numpy(), val.cpu().numpy(), st))
    return out


@pytest.mark.parametrize("case", ["vlad_self_k5", "vlad_self_k100", "queries_vs_db", "partial_chain", "ties_and_near_ties",
                                  "overflow_to_exact", "not_qualified", "three_db_panels", "self_two_panels", "long_rows_self",
                                  "long_rows_queries"])
def test_filtered_topk_is_bit_identical_to_the_exact_path(gpu_ctx, tables, case):
    """pvs_cosine_topk_filtered_dev must return exactly what pvs_cosine_topk_dev returns: indices and fp32 score bits."""
    rng = np.random.default_rng(hash(case) % 2**32)
    same, k = True, 5
    if case.startswith("vlad_self"):
        proto = synth.sift_prototypes()
        raws = [synth.rootsift(synth.sift_like(int(rng.integers(40, 200)), rng, proto).astype(np.float32)) for _ in range(700)]
        packed, off = pack_descriptors(raws, 128)
        q = db = gpu_ctx.vlad_encode(gpu_ctx.codebook(tables["centroids"]), packed, off, DESC_F32)
        k = 100 if case.endswith("k100") else 5
    elif case == "queries_vs_db":
        same = False
        db = rng.standard_normal((1500, 4096)).astype(np.float32)
        q = (db[rng.integers(0, 1500, 300)] + 0.7 * rng.standard_normal((300, 4096))).astype(np.float32)
        k = 10
    elif case == "partial_chain":            # L = 2600: chains of 1024, 1024 and 552
        q = db = (rng.standard_normal((900, 2600)) * rng.uniform(1e-3, 1e3, size=(900, 1))).astype(np.float32)
        k = 7
    elif case == "ties_and_near_ties":
        base = rng.standard_normal((300, 2048)).astype(np.float32)
        dup = base[:100].copy()                                            # exact duplicates: tied scores, index order decides
        near = (base[100:200] * (1 + 1e-4 * rng.standard_normal((100, 2048)))).astype(np.float32)   # inside the margin
        q = db = np.concatenate([base, dup, near, np.zeros((3, 2048), np.float32)])                  # and zero rows
        k = 5
    elif case == "three_db_panels":          # 70000 database rows = 3 score panels: running threshold, appended candidates
        same = False
        db = rng.standard_normal((70000, 256)).astype(np.float32)
        q = (db[rng.integers(0, 70000, 400)] + 0.5 * rng.standard_normal((400, 256))).astype(np.float32)
        k = 8
    elif case == "self_two_panels":          # self-similarity larger than one panel
        q = db = rng.standard_normal((40000, 64)).astype(np.float32)
        k = 5
    elif case == "long_rows_self":           # L = 100,000: the prefilter GEMM runs in four accumulated segments
        q = db = rng.standard_normal((700, 100000)).astype(np.float32)
        q[5] = q[9]
        k = 6
    elif case == "long_rows_queries":
        same = False
        db = rng.standard_normal((900, 70000)).astype(np.float32)
        q = (db[rng.integers(0, 900, 130)] + 0.8 * rng.standard_normal((130, 70000))).astype(np.float32)
        k = 4
    elif case == "overflow_to_exact":        # every column within the margin of every other: more candidates than slots
        c = rng.standard_normal((1, 1024)).astype(np.float32)
        q = db = (c + 1e-3 * rng.standard_normal((600, 1024))).astype(np.float32)
        k = 5
    else:                                    # L % 8 != 0: silently the plain exact path
        q = db = rng.standard_normal((200, 1001)).astype(np.float32)
        k = 5
    (ei, ev, _), (fi, fv, st) = _topk_both(gpu_ctx, np.ascontiguousarray(q), np.ascontiguousarray(db), k, same)
    assert np.array_equal(ei, fi), (case, np.argwhere(ei != fi)[:5])
    assert np.array_equal(ev.view(np.uint32), fv.view(np.uint32)), (case, np.argwhere(ev != fv)[:5])
    if case in ("not_qualified", "overflow_to_exact"):
        assert not st["filtered"]            # (every query overflows its slots: the filter gives up, the exact path runs)
    elif case == "self_two_panels":
        pass                                 # 64-dim random rows: crowded scores, the filter may or may not give up
    else:
        assert st["filtered"] and st["candidates"] >= k * q.shape[0] - 3 * k
        # the three all-zero rows score 0 against everything (all columns tie): they are redone by the exact path
        assert st["redone_exact"] == {"ties_and_near_ties": 3}.get(case, 0), st


@pytest.mark.parametrize("kind", [DESC_F32, DESC_U8_ROOTSIFT, DESC_F32_ROOTSIFT])
def test_prefiltered_assignment_equals_the_exact_kernel(gpu_ctx, tables, kind, monkeypatch):
    """K1 runs an fp16 MFMA prefilter and sends only near ties to the exact fp32 kernel: the labels must be those of the
    exact kernel alone, for every descriptor -- incl. duplicated centres (exact ties), non-finite rows and zero rows."""
    rng = np.random.default_rng(11)
    proto = synth.sift_prototypes()
    raw = synth.sift_like(300000, rng, proto)
    C = tables["centroids"].copy()
    C[17] = C[200]                                   # two identical centres: the first one must win, everywhere
    cb = gpu_ctx.codebook(C)
    if kind == DESC_F32:
        x = synth.rootsift(raw)
        x[5] = 0.0
        x[6, 3] = np.nan
        x[7, 9] = np.inf
        x[8] = C[100]                                 # a descriptor sitting on a centre
        x[9] = 0.5 * (C[3] + C[4])                    # and one between two centres
    elif kind == DESC_F32_ROOTSIFT:
        # raw float rows (non-integer, every scale): the prefilter converts them approximately, the exact kernel exactly
        x = (raw * rng.uniform(1e-3, 1e3, size=(len(raw), 1))).astype(np.float32)
        x[5] = 0.0
        x[6, 3] = np.nan
        x[7, 9] = np.inf
        x[8, 4] = -3.0                                # sqrt of a negative element: NaN in the reference too
        x[9] = 1e-30 * raw[9]                         # quotient denormal
        x[10] = 1e30 * raw[10]
    else:
        x = raw.astype(np.uint8)
        x[5] = 0
        x[6] = 255                                    # the largest row sum
        x[7] = 0
        x[7, 77] = 1                                  # the smallest non-zero row
    off = np.array([0, len(x)], np.int64)
    with gpu_ctx.option(_ffi.OPT_ASSIGN_PREFILTER, 0):
        v_exact, exact = gpu_ctx.vlad_encode(cb, x, off, kind, return_labels=True)
    v_pre, pre = gpu_ctx.vlad_encode(cb, x, off, kind, return_labels=True)
    assert np.array_equal(exact, pre), np.argwhere(exact != pre)[:10]
    assert not np.any(pre == 200)                    # never the later duplicate
    # the measurement variants with two / one fp16 product(s) per (row, cluster) and the wider margin, and (4) the three products on
    # v_mfma_f32_16x16x32_f16 (assign16x_kernel): the same labels and bits
    for variant in (2, 3, 4):
        with gpu_ctx.option(_ffi.OPT_ASSIGN_PREFILTER, variant):
            v_n, lab_n = gpu_ctx.vlad_encode(cb, x, off, kind, return_labels=True)
        assert np.array_equal(exact, lab_n), (variant, np.argwhere(exact != lab_n)[:10])
        assert v_exact.tobytes() == v_n.tobytes()
    # the encodings too, bit for bit: with the prefilter the aggregate pass of uint8 rows converts elements from the row
    # statistics the prefilter leaves behind, without it from its own row reduction
    assert v_exact.tobytes() == v_pre.tobytes()


@pytest.mark.parametrize("kind", [DESC_F32, DESC_U8_ROOTSIFT])
@pytest.mark.parametrize("rows", [(3, 40, 512, 0, 700, 64), (4100, 9000)])
def test_aggregate_variants_give_the_same_bits(gpu_ctx, tables, kind, rows):
    """The gather aggregate has two register budgets at D <= 128 (eight waves per SIMD with batches of 4 rows for short images,
    five waves with batches of 8 for long ones; the launcher picks by rows per cluster).  Same arithmetic in the same order:
    encodings, labels and 1 / norm must be bit-identical whichever runs, on short and on long images."""
    rng = np.random.default_rng(5)
    raws = [synth.sift_like(n, rng) for n in rows]
    cb = gpu_ctx.codebook(tables["centroids"])
    imgs = [r.astype(np.uint8) for r in raws] if kind == DESC_U8_ROOTSIFT else [synth.rootsift(r) for r in raws]
    packed, off = pack_descriptors(imgs, 128, np.uint8 if kind == DESC_U8_ROOTSIFT else np.float32)
    outs = []
    for variant in (1, 2, 0):
        with gpu_ctx.option(_ffi.OPT_AGG_VARIANT, variant):
            v, lab = gpu_ctx.vlad_encode(cb, packed, off, kind, return_labels=True)
        outs.append((v.tobytes(), lab.tobytes()))
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("K,D", [(40, 100), (64, 30), (256, 128), (17, 16), (130, 72)])
def test_prefiltered_assignment_odd_shapes(gpu_ctx, K, D, monkeypatch):
    """Prefilter vs exact kernel on shapes that pad (K to 32 NT, D to 16) and on the scalar-load path (D % 4 != 0)."""
    rng = np.random.default_rng(K * 1000 + D)
    C = rng.normal(0.0, 1.0, size=(K, D)).astype(np.float32)
    x = (C[rng.integers(0, K, 20000)] + rng.normal(0.0, 0.8, size=(20000, D))).astype(np.float32)
    x[:50] *= 1e4                                   # rows far outside the table's scale
    x[50:100] *= 1e-4
    cb = gpu_ctx.codebook(C)
    off = np.array([0, len(x)], np.int64)
    with gpu_ctx.option(_ffi.OPT_ASSIGN_PREFILTER, 0):
        _, exact = gpu_ctx.vlad_encode(cb, x, off, DESC_F32, return_labels=True)
    _, pre = gpu_ctx.vlad_encode(cb, x, off, DESC_F32, return_labels=True)
    assert np.array_equal(exact, pre), np.argwhere(exact != pre)[:10]
    assert np.array_equal(pre, orc.kmeans_predict(x, C)) or np.mean(pre != orc.kmeans_predict(x, C)) < 1e-3


@pytest.mark.parametrize("M,N,L", [(300, 300, 32768), (130, 257, 4096), (129, 700, 2600), (64, 64, 40)])
def test_cosine_scores_are_the_defined_fp32_recurrence(gpu_ctx, M, N, L):
    """Beyond the 2e-6 tolerance against the reference's BLAS result: the exact GEMM's score is a defined recurrence
    (fma chain in the MFMA's k order, chains of 1024, ordered chain sums) that the C checker reproduces bit for bit --
    in the symmetric, general, split-K and short-row (L < 1024, L % 32 != 0) cases alike."""
    import pvsim_oracle_c as orc_c
    rng = np.random.default_rng(M + N + L)
    a = rng.standard_normal((M, L)).astype(np.float32)
    same = M == N
    b = a if same else rng.standard_normal((N, L)).astype(np.float32)
    pairs = np.stack([rng.integers(0, M, 400), rng.integers(0, N, 400)], axis=1)
    # unit norms through the raw GEMM entry point: the 1/||row|| factors come from a separate kernel with its own order
    import torch
    dev = torch.device("cuda", 0)
    ta, tb = torch.from_numpy(a).to(dev), (None if same else torch.from_numpy(b).to(dev))
    tb = ta if same else tb
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.cosine_dev(ta.data_ptr(), M, tb.data_ptr(), N, L, None, None, out.data_ptr(), N)
    gpu_ctx.sync()
    raw = out.cpu().numpy()
    want = orc_c.cosine_chain(a, b, None, None, pairs)
    assert np.array_equal(raw[pairs[:, 0], pairs[:, 1]].view(np.uint32), want.view(np.uint32))


def test_notebook_shapes_known_answers():
    """The reference has no tests; its executed notebooks record output shapes:
    getting_started: VLAD K=32 on PCA-64 SIFT -> (N, 2048), Fisher -> (N, 4128) = 32 + 2*32*64;
    pipeline: VGG16 deep features (514-D incl. coordinates): VLAD K=256 -> (1, 131584) = 256*514, Fisher on the PCA-257
    features -> (1, 131840) = 256 + 2*256*257, Pipeline([VLAD, Fisher]) -> (5, 263424)."""
    from pvsim.encoders import VLADEncoder, FisherVectorEncoder, Pipeline
    from pvsim.features import Lambda
    from pvsim.models import KMeansModel, GMMModel, PCAModel
    rng = np.random.default_rng(4)
    sift = Lambda(lambda im: im.astype(np.float32), 128)
    imgs = [rng.integers(0, 255, size=(int(n), 128)) for n in (60, 75, 90, 64, 80)]
    pca64 = PCAModel(np.linalg.qr(rng.standard_normal((128, 64)))[0].T, rng.random(128))
    v = VLADEncoder(sift, kmeans_model=KMeansModel(rng.random((32, 64))), pca=pca64).encode(imgs)
    f = FisherVectorEncoder(sift, gmm_model=GMMModel(np.full(32, 1 / 32), rng.random((32, 64)), np.ones((32, 64))), pca=pca64).encode(imgs)
    assert v.shape == (5, 2048) and v.dtype == np.float32
    assert f.shape == (5, 4128) and f.dtype == np.float64
    deep = Lambda(lambda im: im.astype(np.float32) / 64.0, 514)
    dimgs = [rng.integers(0, 255, size=(196, 514)) for _ in range(5)]
    pca257 = PCAModel(np.linalg.qr(rng.standard_normal((514, 257)))[0].T, rng.random(514))
    ve = VLADEncoder(deep, kmeans_model=KMeansModel(rng.random((256, 514))))
    fe = FisherVectorEncoder(deep, gmm_model=GMMModel(np.full(256, 1 / 256), rng.random((256, 257)), np.ones((256, 257))), pca=pca257)
    assert ve.encode(dimgs[:1]).shape == (1, 131584)
    assert fe.encode(dimgs[:1]).shape == (1, 131840)
    p = Pipeline([ve, fe]).encode(dimgs)
    assert p.shape == (5, 263424)
    s = ve.similarity_score(dimgs[:2], dimgs[2:])
    assert s.shape == (2, 3) and s.dtype == np.float32


def test_config2_full_size_topk_against_the_restatement(gpu_ctx, tables):
    """SURVEY.md section 8d, config 2 at its full size: 8189 images (the bench generator, on the device), VLAD K=256,
    8189 x 8189 retrieval with k = 5 and k = 100, compared with the NumPy restatement (sklearn normalize + BLAS product +
    stable ranking) on a 512-query subsample.  Adjacent scores at rank ~100 are ~6e-5 apart, so an fp32 evaluation in another summation
    order may swap near ties: ranks are compared through their scores (3e-6), the k = 5 lists exactly."""
    import sys
    import torch
    from conftest import REPO
    sys.path.insert(0, REPO)
    import bench
    dev = torch.device("cuda", 0)
    N, L = 8189, 32768
    raw, offsets = bench.make_corpus(N, 1235, dev)
    d_off = torch.from_numpy(offsets).to(dev)
    enc = torch.empty((N, L), dtype=torch.float32, device=dev)
    inv = torch.empty((N,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    cb = gpu_ctx.codebook(tables["centroids"])
    gpu_ctx.vlad_encode_dev(cb, raw.data_ptr(), DESC_U8_ROOTSIFT, d_off.data_ptr(), N, int(offsets[-1]), enc.data_ptr(),
                            d_inv_norm=inv.data_ptr())
    gpu_ctx.sync()
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(N, 512, replace=False))
    h_enc = enc.cpu().numpy()
    sims = orc.cosine_similarity(h_enc[sub], h_enc)
    for k in (5, 100):
        idx = torch.empty((N, k), dtype=torch.int64, device=dev)
        val = torch.empty((N, k), dtype=torch.float32, device=dev)
        fidx, fval = torch.empty_like(idx), torch.empty_like(val)
        torch.cuda.synchronize()
        gpu_ctx.cosine_topk_dev(enc.data_ptr(), N, enc.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), k, 0, False,
                                idx.data_ptr(), val.data_ptr())
        st = gpu_ctx.cosine_topk_filtered_dev(enc.data_ptr(), N, enc.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), k,
                                              fidx.data_ptr(), fval.data_ptr())
        gpu_ctx.sync()
        assert st["filtered"] and torch.equal(idx, fidx) and torch.equal(val.view(torch.int32), fval.view(torch.int32))
        ridx, rval = orc.topk(sims, k)
        di, dv = idx.cpu().numpy()[sub], val.cpu().numpy()[sub]
        np.testing.assert_allclose(dv, rval, rtol=0, atol=3e-6)   # NumPy's own self-scores reach 1.000001..1.000003 at L = 32768
        if k == 5:
            assert np.array_equal(di, ridx)
        else:
            agree = (di == ridx).mean()
            assert agree > 0.99, agree                      # the rest are swaps of near ties (scores equal to 3e-6, above)
        assert np.array_equal(di[:, 0], sub)


def test_learn_from_raw_sift_descriptors(tables):
    """learn_from_descriptors(rootsift=True): raw uint8 SIFT rows are transformed on the device exactly as encode does, so a
    vocabulary learnt from them equals one learnt from the host-side RootSIFT rows (same start, same iterations)."""
    from pvsim.encoders import VLADEncoder
    from pvsim.features import Lambda
    from pvsim.models import KMeansModel
    rng = np.random.default_rng(8)
    raw = synth.sift_like(12000, rng).astype(np.uint8)
    x = synth.rootsift(raw.astype(np.float32))
    c0 = x[rng.choice(len(x), 32, replace=False)]
    fx = Lambda(lambda im: im.astype(np.float32), 128)
    a = VLADEncoder(fx, kmeans_model=KMeansModel(c0))
    b = VLADEncoder(fx, kmeans_model=KMeansModel(c0))
    a.learn_from_descriptors(raw, n_clusters=32, rootsift=True, init=c0, n_init=1, max_iter=4, tol=0.0)
    b.learn_from_descriptors(x, n_clusters=32, init=c0, n_init=1, max_iter=4, tol=0.0)
    assert np.array_equal(a.clustering_model.cluster_centers_, b.clustering_model.cluster_centers_)
    assert np.array_equal(a.clustering_model.labels_, b.clustering_model.labels_)
    ref_c, ref_l, _, _ = orc.kmeans_lloyd(x, c0, max_iter=4, tol=0.0)
    assert np.mean(ref_l != a.clustering_model.labels_) < 2e-3          # near ties may flip (fp32 sums in another order)
    np.testing.assert_allclose(a.clustering_model.cluster_centers_, ref_c, rtol=0, atol=2e-4)


def test_learn_gmm_more_than_256_components(gpu_ctx):
    """EM with K = 300: the posterior runs on the general kernel, the moments 256 components at a time; three iterations
    against the NumPy restatement of scikit-learn's loop."""
    import warnings
    from pvsim import learn
    rng = np.random.default_rng(12)
    K, D, n = 300, 24, 30000
    mu = rng.normal(0, 4, (K, D))
    x = (mu[rng.integers(0, K, n)] + rng.standard_normal((n, D))).astype(np.float32)
    rows = learn.DeviceRows.from_host(gpu_ctx, x)
    w0, m0, p0 = np.full(K, 1.0 / K), mu + 0.3 * rng.standard_normal((K, D)), np.ones((K, D))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g = learn.fit_gmm(rows, K, weights_init=w0, means_init=m0, precisions_init=p0, max_iter=3, tol=0.0)
    w, m, c, lower, _, _ = orc.gmm_em(x, w0, m0, 1.0 / p0, max_iter=3, tol=0.0)
    np.testing.assert_allclose(g.weights_, w, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(g.means_, m, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(g.covariances_, c, rtol=1e-8, atol=1e-10)
    assert abs(g.lower_bound_ - lower) < 1e-9
    rows.free()


@pytest.mark.parametrize("nq,N,k", [(9, 8189, 5), (4, 300, 16), (3, 20000, 1), (6, 7, 10), (5, 40000, 8), (7, 70000, 10), (3, 16384, 16)])
def test_small_k_kernel_equals_the_radix_select(gpu_ctx, nq, N, k, monkeypatch):
    """k <= 16 takes the extraction kernel: same keys, same order as the radix select -- ties, NaN, -0.0, -inf, fewer columns
    than k, several 8192-column chunks and the running-list merge included."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(nq * 1000 + N + k)
    s = rng.standard_normal((nq, N)).astype(np.float32)
    m = s[:, 1::7].shape[1]
    s[:, 0:7 * m:7] = s[:, 1::7]                                # exact ties
    s[0, :3] = np.nan
    s[1, 2] = -0.0
    s[1, 3] = 0.0
    s[2, :] = -np.inf
    panels = [(0, N // 2), (N // 2, N)] if N > 20 else [(0, N)]
    out = []
    for select_only in (1, 2, 3, 0):       # radix select, k rounds, threshold filter + rounds, the launcher's own choice
        gpu_ctx.set_option(_ffi.OPT_TOPK_SELECT_ONLY, int(select_only))
        idx = torch.full((nq, k), -9, dtype=torch.int64, device=dev)
        val = torch.full((nq, k), -9.0, dtype=torch.float32, device=dev)
        for p, (c0, c1) in enumerate(panels):                  # second panel merges into the running lists
            t = torch.from_numpy(np.ascontiguousarray(s[:, c0:c1])).to(dev)
            torch.cuda.synchronize()
            gpu_ctx.topk_dev(t.data_ptr(), nq, c1 - c0, c1 - c0, k, c0, p > 0, idx.data_ptr(), val.data_ptr())
            gpu_ctx.sync()
        out.append((idx.cpu().numpy(), val.cpu().numpy()))
    gpu_ctx.set_option(_ffi.OPT_TOPK_SELECT_ONLY, 0)
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0])
        assert np.array_equal(out[0][1].view(np.uint32), o[1].view(np.uint32))


def test_u8_rootsift_short_sequence_equals_ieee(gpu_ctx):
    """uint8 rows take a shorter RootSIFT sequence (one reciprocal per row, fma-refined quotient and rsq-based root;
    desc_load.hpp:RootsiftRow) that must give NumPy's bits for every (element, row sum) pair.  csrc/bench/rootsift_exhaustive.hip
    checks all 8.3 M pairs on the device; here: 3*10^5 rows of every density through the C-ABI against the oracle, including the
    one-element rows (element = sum), the all-255 row and the empty row."""
    import torch
    rng = np.random.default_rng(5)
    blocks = []
    for hi, p in [(256, 1.0), (256, 0.5), (256, 0.1), (32, 0.7), (4, 0.3), (256, 0.02), (2, 0.5), (120, 0.9)]:
        b = rng.integers(0, hi, (37500, 128), dtype=np.uint8)
        b[rng.random(b.shape) > p] = 0
        blocks.append(b)
    single = np.zeros((256, 128), np.uint8)
    single[np.arange(256), rng.integers(0, 128, 256)] = np.arange(256)
    blocks += [single, np.full((1, 128), 255, np.uint8), np.zeros((1, 128), np.uint8)]
    raw = np.ascontiguousarray(np.concatenate(blocks))
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw).to(dev)
    d_out = torch.empty(raw.shape, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.materialise_dev(d_raw.data_ptr(), DESC_U8_ROOTSIFT, 128, raw.shape[0], d_out.data_ptr())
    gpu_ctx.sync()
    want = orc.rootsift(raw)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------------------------------- fused one-read encode (round 2)
@pytest.mark.gpu
@pytest.mark.parametrize("kind", [DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT])
def test_fused_encode_equals_the_two_kernel_path(gpu_ctx, tables, kind):
    """vlad_fused_kernel (every descriptor read once: prefilter scores, exact re-evaluation of near ties, residual sums in
    descriptor order and normalisation in one persistent kernel) against assign + gather aggregate: labels, encodings and
    1/||row|| bit for bit, on ragged images (empty, 1, 63, 64, 65, several stages, > 4096 rows) with rows the prefilter
    cannot settle or cannot take (copies of centres, midpoints, zero rows, NaN / inf, far out of the table's scale)."""
    import torch
    C = tables["centroids"]
    rng = np.random.default_rng(77 + kind)
    counts = [0, 1, 63, 64, 65, 130, 512, 700, 0, 4100, 257, 31, 1411, 2, 199]
    raws = [synth.sift_like(n, rng) for n in counts]
    if kind == DESC_F32:
        imgs = [synth.rootsift(r) if len(r) else np.zeros((0, 128), np.float32) for r in raws]
        x = imgs[6]
        x[5] = 0.0
        x[6, 3] = np.nan
        x[7, 9] = np.inf
        x[8] = C[100]
        x[9] = 0.5 * (C[3] + C[4])
        x[10] *= 1e6
        x[11] *= 1e-6
        x[12] *= 1e30
        x[13] = -x[13]
        imgs[9][4000:4100] = 0.5 * (C[rng.integers(0, 256, 100)] + C[rng.integers(0, 256, 100)])   # near ties, last stage
        dt = np.float32
    elif kind == DESC_F32_ROOTSIFT:
        imgs = [r.astype(np.float32) for r in raws]
        imgs[6][5] = 0.0
        dt = np.float32
    else:
        imgs = [r.astype(np.uint8) for r in raws]
        imgs[6][5] = 0
        dt = np.uint8
    packed, off = pack_descriptors(imgs, 128, dt)
    cb = gpu_ctx.codebook(C)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(packed).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    n, total = len(imgs), int(off[-1])
    res = {}
    for name, path in (("gather", _ffi.VLAD_PATH_GATHER), ("fused", _ffi.VLAD_PATH_FUSED)):
        out = torch.full((n, 256 * 128), 7.0, dtype=torch.float32, device=dev)
        lab = torch.full((total,), -5, dtype=torch.int32, device=dev)
        inv = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        with gpu_ctx.option(_ffi.OPT_VLAD_PATH, path):
            gpu_ctx.vlad_encode_dev(cb, d_x.data_ptr(), kind, d_off.data_ptr(), n, total, out.data_ptr(), d_labels=lab.data_ptr(),
                                    d_inv_norm=inv.data_ptr())
            gpu_ctx.sync()
        res[name] = (out.cpu().numpy(), lab.cpu().numpy(), inv.cpu().numpy())
    (o1, l1, i1), (o2, l2, i2) = res["gather"], res["fused"]
    assert np.array_equal(l1, l2), np.argwhere(l1 != l2)[:10]
    assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32)), np.argwhere(o1.view(np.uint32) != o2.view(np.uint32))[:10]
    assert np.array_equal(i1.view(np.uint32), i2.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("power,order,eps", [(0.5, 2, 1e-9), (1.0, 1, 1e-9), (0.3, np.inf, 1e-6), (1.0, 3.0, 0.0)])
def test_fused_encode_normalisation_variants(gpu_ctx, tables, power, order, eps):
    """The fused kernel's K3 epilogue in every normalisation mode, bit for bit against the gather kernel's epilogue."""
    rng = np.random.default_rng(5)
    imgs = [synth.rootsift(synth.sift_like(n, rng)) for n in (40, 300, 1, 129)]
    packed, off = pack_descriptors(imgs, 128, np.float32)
    cb = gpu_ctx.codebook(tables["centroids"])
    with gpu_ctx.option(_ffi.OPT_VLAD_PATH, _ffi.VLAD_PATH_GATHER):
        a = gpu_ctx.vlad_encode(cb, packed, off, DESC_F32, power, order, eps)
    with gpu_ctx.option(_ffi.OPT_VLAD_PATH, _ffi.VLAD_PATH_FUSED):
        b = gpu_ctx.vlad_encode(cb, packed, off, DESC_F32, power, order, eps)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_fused_encode_labels_equal_the_exact_kernel_on_many_rows(gpu_ctx, tables):
    """2 M RootSIFT-like rows plus 2e5 engineered near ties: the fused kernel's labels against the exact f32 MFMA kernel's
    (PVS_OPT_ASSIGN_PREFILTER = 0)."""
    import torch
    C = tables["centroids"]
    rng = np.random.default_rng(2024)
    dev = torch.device("cuda", 0)
    x = synth.rootsift(synth.sift_like(2_000_000, rng))
    a_, b_ = rng.integers(0, 256, 200_000), rng.integers(0, 256, 200_000)
    t = rng.uniform(0.4999, 0.5001, size=(200_000, 1)).astype(np.float32)
    ties = (t * C[a_] + (1 - t) * C[b_]).astype(np.float32)
    x = np.concatenate([x, ties])
    n_img = 1100
    off = np.linspace(0, len(x), n_img + 1).astype(np.int64)
    cb = gpu_ctx.codebook(C)
    d_x = torch.from_numpy(x).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    out = torch.empty((n_img, 256 * 128), dtype=torch.float32, device=dev)
    labs = []
    for pre in (0, 1):
        lab = torch.full((len(x),), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        with gpu_ctx.option(_ffi.OPT_ASSIGN_PREFILTER, pre), gpu_ctx.option(_ffi.OPT_VLAD_PATH, _ffi.VLAD_PATH_FUSED if pre else _ffi.VLAD_PATH_GATHER):
            gpu_ctx.vlad_encode_dev(cb, d_x.data_ptr(), DESC_F32, d_off.data_ptr(), n_img, len(x), out.data_ptr(), d_labels=lab.data_ptr())
            gpu_ctx.sync()
        labs.append(lab.cpu().numpy())
    assert np.array_equal(labs[0], labs[1]), (np.argwhere(labs[0] != labs[1])[:10], int((labs[0] != labs[1]).sum()))


def test_short_sqrt_and_division_are_the_ieee_results():
    """The VLAD normalisation epilogue takes its square root and its division by the cluster norm from short instruction
    sequences (desc_load.hpp: sqrt_rn, DivByRow).  The device-side checker compares them with the compiler's IEEE sqrtf / division:
    every one of the 2^32 float bit patterns for the root; 2e10 dividend / divisor pairs incl. the edge patterns, the guarded
    fallbacks and the unguarded call of the epilogue for the division.  Bit for bit."""
    import os, subprocess
    from conftest import REPO
    csrc = os.path.join(REPO, "python-visual-similarity_amd", "csrc")
    exe = os.path.join(csrc, "bench", "exact_sqrt_div")
    if not os.path.exists(exe):          # normally built by `make` in csrc (__graft_entry__.build); same toolchain on the GPU box
        subprocess.run(["make", "-C", csrc, "bench/exact_sqrt_div"], check=True, capture_output=True, timeout=600)
    assert os.path.exists(exe)
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL EQUAL" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "over all 2^32 bit patterns: 0 differ" in r.stdout
