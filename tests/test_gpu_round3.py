"""Round-3 GPU parity tests (through the C-ABI): the float64 similarity on the f64 matrix pipe and its device-resident entry
points -- the reference's dtype rule for Fisher encodings (pyvisim/_utils.py:312-330: float32 only when BOTH operands are
float32; eval.py:37-43 ranks that array)."""
import numpy as np
import pytest

import pvsim_oracle as orc

pytestmark = pytest.mark.gpu


def _dev_f64(ctx, a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return ctx.buffer(max(a.nbytes, 16)).upload(a)


# shapes: one tile with edges / several tiles, general order / odd L (vector-ALU tile kernel) / a k-tile tail (L % 16 != 0) /
# enough tiles for full rounds plus a split-K tail / L shorter than one k-tile
@pytest.mark.parametrize("M,N,L", [(1, 7, 6), (130, 257, 1026), (64, 64, 15), (300, 129, 4100), (1200, 9000, 258), (5, 3, 2), (640, 8200, 64)])
def test_f64_cosine_dev_against_numpy(gpu_ctx, M, N, L):
    """pvs_cosine_f64_dev (v_mfma_f64_16x16x4_f64 when rows are 16-B aligned) = NumPy's float64 cosine to 1e-13."""
    rng = np.random.default_rng(M * 7 + N * 3 + L)
    a = rng.standard_normal((M, L)) * np.exp(rng.uniform(-3, 3, size=(M, 1)))
    b = rng.standard_normal((N, L))
    if N > 4:
        b[3] = 0.0                                     # zero row: norm treated as 1, scores exactly 0
    da, db = _dev_f64(gpu_ctx, a), _dev_f64(gpu_ctx, b)
    ia, ib = gpu_ctx.buffer(M * 8), gpu_ctx.buffer(N * 8)
    gpu_ctx.row_inv_norms_f64_dev(da.ptr, M, L, ia.ptr)
    gpu_ctx.row_inv_norms_f64_dev(db.ptr, N, L, ib.ptr)
    out = gpu_ctx.buffer(M * N * 8).fill_bytes(0xff)
    gpu_ctx.cosine_f64_dev(da.ptr, M, db.ptr, N, L, ia.ptr, ib.ptr, out.ptr, N)
    gpu_ctx.sync()
    got = out.download((M, N), np.float64)
    ref = orc.cosine_similarity(a, b)
    assert ref.dtype == np.float64 and np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13)
    if N > 4:
        assert (got[:, 3] == 0).all()


@pytest.mark.parametrize("N,L", [(300, 4098), (1000, 130), (2949, 64), (8189, 34)])
def test_f64_self_similarity_is_bitwise_symmetric(gpu_ctx, N, L):
    """A == B: only the upper triangle of 128 x 128 tiles is computed, the rest mirrored (and, for few tiles or a partly filled
    last round, tiles are cut along k: 2949 rows = 24 x 24 tiles -> every tile split; 8189 rows -> 4 full rounds + a split tail).
    out == out.T bit for bit, the diagonal is 1 to 1e-15, everything equals NumPy to 1e-13."""
    rng = np.random.default_rng(N + L)
    a = rng.standard_normal((N, L))
    a[1::9] = a[0::9][: len(a[1::9])]                 # duplicated rows
    da = _dev_f64(gpu_ctx, a)
    ia = gpu_ctx.buffer(N * 8)
    gpu_ctx.row_inv_norms_f64_dev(da.ptr, N, L, ia.ptr)
    out = gpu_ctx.buffer(N * N * 8).fill_bytes(0xff)
    gpu_ctx.cosine_f64_dev(da.ptr, N, da.ptr, N, L, ia.ptr, ia.ptr, out.ptr, N)
    gpu_ctx.sync()
    got = out.download((N, N), np.float64)
    assert np.array_equal(got, got.T)
    np.testing.assert_allclose(np.diag(got), 1.0, rtol=0, atol=4e-15)
    rows = rng.choice(N, size=min(N, 64), replace=False)
    np.testing.assert_allclose(got[rows], orc.cosine_similarity(a[rows], a), rtol=0, atol=1e-13)
    # the host-pointer form of the same product (pvs_cosine, is_f64) goes through the same kernels
    if N <= 1000:
        np.testing.assert_array_equal(gpu_ctx.cosine(a, a), got)


@pytest.mark.parametrize("nq,N,L,k", [(7, 500, 66, 5), (300, 300, 130, 300), (40, 9000, 18, 9000), (9000, 9000, 18, 3)])
def test_f64_topk_dev_equals_host_form_and_stable_argsort(gpu_ctx, nq, N, L, k):
    """pvs_cosine_topk_f64_dev (operands, norms, panels and lists resident) = pvs_cosine_topk_f64 (host pointers, one upload) =
    a stable argsort of the device's own float64 scores; incl. Q == DB (symmetric kernel) and a full-depth ranking of 9000 rows."""
    rng = np.random.default_rng(nq + N + L + k)
    db = rng.standard_normal((N, L))
    db[2::11] = db[0::11][: len(db[2::11])]
    same = nq == N
    q = db if same else rng.standard_normal((nq, L))
    ddb = _dev_f64(gpu_ctx, db)
    dq = ddb if same else _dev_f64(gpu_ctx, q)
    idb = gpu_ctx.buffer(N * 8)
    gpu_ctx.row_inv_norms_f64_dev(ddb.ptr, N, L, idb.ptr)
    iq = idb if same else gpu_ctx.buffer(nq * 8)
    if not same:
        gpu_ctx.row_inv_norms_f64_dev(dq.ptr, nq, L, iq.ptr)
    d_idx, d_val = gpu_ctx.buffer(nq * k * 8), gpu_ctx.buffer(nq * k * 8)
    gpu_ctx.cosine_topk_f64_dev(dq.ptr, nq, ddb.ptr, N, L, iq.ptr, idb.ptr, k, d_idx.ptr, d_val.ptr)
    gpu_ctx.sync()
    idx, val = d_idx.download((nq, k), np.int64), d_val.download((nq, k), np.float64)
    h_idx, h_val = gpu_ctx.cosine_topk_f64(q, db, k)
    assert np.array_equal(idx, h_idx) and np.array_equal(val, h_val)
    sub = slice(0, min(nq, 48))
    full = gpu_ctx.cosine(q[sub], db) if not same else None
    if full is None:                                   # the symmetric kernel's scores: take them from the device product itself
        out = gpu_ctx.buffer(N * N * 8)
        gpu_ctx.cosine_f64_dev(ddb.ptr, N, ddb.ptr, N, L, idb.ptr, idb.ptr, out.ptr, N)
        gpu_ctx.sync()
        full = out.download((N, N), np.float64)[sub]
    np.testing.assert_allclose(full, orc.cosine_similarity(q[sub], db), rtol=0, atol=1e-13)
    order = np.argsort(-full, axis=1, kind="stable")[:, :k]
    assert np.array_equal(idx[sub], order)
    assert np.array_equal(val[sub], np.take_along_axis(full, order, 1))


def test_top_k_map_full_depth_on_a_float64_database_larger_than_8192():
    """eval.top_k_map(k=None) with float64 encodings and N > 8192 (used to raise NotImplementedError): the lists are the
    oracle's full argsort wherever the oracle's scores are not tied within 1e-12."""
    from pvsim import eval as pe
    rng = np.random.default_rng(5)
    N, L, nq = 8300, 24, 6
    db = rng.standard_normal((N, L))
    q = db[:nq] + 0.1 * rng.standard_normal((nq, L))
    idx, val = pe._rank(q, db, None)
    assert idx.shape == (nq, N) and val.dtype == np.float64
    s = orc.cosine_similarity(q, db)
    ref = np.argsort(-s, axis=1, kind="stable")
    sv = np.take_along_axis(s, ref, 1)
    np.testing.assert_allclose(val, sv, rtol=0, atol=1e-13)
    clear = np.ones_like(ref, dtype=bool)
    gap = np.abs(np.diff(sv, axis=1)) > 1e-12
    clear[:, 1:] &= gap
    clear[:, :-1] &= gap
    assert np.array_equal(idx[clear], ref[clear])
    assert (np.sort(idx, axis=1) == np.arange(N)).all()          # a permutation: every row ranked exactly once


# ======================================================================================= the reference's shipped vocabularies
def test_shipped_vocabularies_through_the_class_api():
    """`FisherVectorEncoder(weights=GMMWeights.*)` / `VLADEncoder(weights=KMeansWeights.*)` as a pyvisim user writes them
    (pyvisim/encoders/_base_encoder.py:117-155, 207-209), on the tables read out of the reference's own model files, against the
    REFERENCE's outputs on those tables (tests/golden/shipped_tables.npz):
      * OXFORD102_K256_ROOTSIFT: 487 covariance entries at the reg_covar floor (precision 1e6) -- the numerically hardest real
        case (cancelling terms of ~1e6 in the log-density); 1e-9 abs on unit-norm float64 vectors;
      * OXFORD102_K256_ROOTSIFT_PCA (128 -> 64) and OXFORD102_K256_VGG16_PCA (514 -> 257; FV length 131,840 as in
        examples/pipeline.ipynb): the PCA member is paired automatically; fp32 projection, tolerance of its summation order;
      * KMeansWeights.OXFORD102_K256_ROOTSIFT: the means_-derived stand-in codebook (warns): labels exact, values 5e-7."""
    from shipped_inputs import shipped_inputs
    from pvsim import synth
    from pvsim.encoders import FisherVectorEncoder, VLADEncoder, GMMWeights, KMeansWeights
    from pvsim.features import Lambda
    from conftest import load_golden
    g = load_golden("shipped_tables")
    raws, deep = shipped_inputs()
    imgs = [r.astype(np.float32) for r in raws]
    rootsift_x = Lambda(synth.rootsift, 128)

    f = FisherVectorEncoder(rootsift_x, weights=GMMWeights.OXFORD102_K256_ROOTSIFT)
    assert f.pca is None and f.clustering_model.means_.shape == (256, 128)
    F = f.encode(imgs)
    assert F.dtype == np.float64 and F.shape == (5, 65792)
    np.testing.assert_allclose(F[0], g["fisher_rootsift_img0"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(F[:, ::16], g["fisher_rootsift_every16"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(f.similarity_score(imgs, imgs), g["fisher_rootsift_cos"].astype(np.float32), rtol=0, atol=1e-6)

    fp = FisherVectorEncoder(rootsift_x, weights=GMMWeights.OXFORD102_K256_ROOTSIFT_PCA)
    assert fp.pca is not None and fp.pca.n_components == 64
    Fp = fp.encode(imgs)
    np.testing.assert_allclose(Fp[0], g["fisher_rootsift_pca_img0"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(Fp[:, ::16], g["fisher_rootsift_pca_every16"], rtol=0, atol=1e-7)

    store = {i: d for i, d in enumerate(deep)}
    fd = FisherVectorEncoder(Lambda(lambda im: store[int(im[0, 0])], 514), weights=GMMWeights.OXFORD102_K256_VGG16_PCA)
    assert fd.pca.n_components == 257 and fd.pca.n_features_in_ == 514
    Fd = fd.encode([np.array([[i]], dtype=np.int64) for i in range(len(deep))])
    assert Fd.shape == (4, 131840)
    np.testing.assert_allclose(Fd[0], g["fisher_vgg16_pca_img0"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(Fd[:, ::16], g["fisher_vgg16_pca_every16"], rtol=0, atol=2e-6)

    with pytest.warns(UserWarning, match="absent from its checkout"):
        v = VLADEncoder(rootsift_x, weights=KMeansWeights.OXFORD102_K256_ROOTSIFT)
    V = v.encode(imgs)
    np.testing.assert_allclose(V, g["vlad_rootsift"], rtol=0, atol=5e-7)
    with pytest.raises(FileNotFoundError):
        GMMWeights.OXFORD102_K256_VGG16.load()        # absent from the reference's checkout too (.MISSING_LARGE_BLOBS:2)


# ======================================================================================= the phase-scheduled fp16 GEMMs at ragged shapes
@pytest.mark.parametrize("M,N,L", [(2100, 33000, 200), (1300, 52000, 136), (4100, 16500, 72)])
def test_fp16_8phase_gemm_on_ragged_shapes(gpu_ctx, M, N, L):
    """pvs_cosine_f16_dev on problems of more than four rounds of 256 x 256 tiles, so that the full rounds run on the 8-phase
    kernel (gemm_f16_8ph.hpp) and the partly filled last round on the split-K two-stage kernel: rows / columns that do not fill
    the edge tiles, L = 200 (three full k-tiles + a partial one: chunks past L staged from zeros, an even tile count), L = 136 (an
    odd tile count: the second buffer of the last iteration is a zero tile), L = 72 (shorter than the prologue's look-ahead).
    Reference: float64 dot products of the fp16-rounded operands."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(M + N + L)
    a = torch.randn((M, L), generator=g, device=dev) * 0.3
    b = torch.randn((N, L), generator=g, device=dev) * 0.3
    b[5] = 0.0
    a16, b16 = a.half().contiguous(), b.half().contiguous()
    ia = (1.0 / a16.float().norm(dim=1)).contiguous()
    ib = (1.0 / b16.float().norm(dim=1).clamp_min(1e-30)).contiguous()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.cosine_f16_dev(a16.data_ptr(), M, b16.data_ptr(), N, L, ia.data_ptr(), ib.data_ptr(), out.data_ptr(), N)
    gpu_ctx.sync()
    ref = (a16.double() @ b16.double().T) * ia.double()[:, None] * ib.double()[None, :]
    err = (out.double() - ref).abs().max().item()
    assert torch.isfinite(out).all() and err < 2e-6, err
    assert (out[:, 5] == 0).all()


@pytest.mark.parametrize("nq,N,L,k", [(2100, 33000, 200, 10), (1100, 70000, 264, 5)])
def test_filtered_topk_through_the_256x128_prefilter_kernel(gpu_ctx, nq, N, L, k):
    """pvs_cosine_topk_filtered_dev on a query block x corpus large enough for the two-level 256 x 128 prefilter GEMM
    (gemm_f16_2lvl.hpp: >= 4 rounds of tiles, general order), ragged in rows, columns and k-tiles (L = 200: partial last k-tile;
    L = 264: five k-tiles, a count that is not a multiple of the three LDS buffers) and over more than one 32768-column panel:
    indices AND float32 score bits identical to the exact path (f32 MFMA GEMM over all pairs + select)."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(nq + N + L + k)
    proto = torch.randn((300, L), generator=g, device=dev)
    db = (proto[torch.randint(0, 300, (N,), generator=g, device=dev)] + 0.6 * torch.randn((N, L), generator=g, device=dev)).contiguous()
    q = (db[torch.randint(0, N, (nq,), generator=g, device=dev)] + 0.3 * torch.randn((nq, L), generator=g, device=dev)).contiguous()
    iq = torch.empty((nq,), dtype=torch.float32, device=dev)
    idb = torch.empty((N,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.row_inv_norms_dev(q.data_ptr(), nq, L, iq.data_ptr())
    gpu_ctx.row_inv_norms_dev(db.data_ptr(), N, L, idb.data_ptr())
    res = []
    for filtered in (False, True):
        idx = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
        val = torch.full((nq, k), -7.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        if filtered:
            st = gpu_ctx.cosine_topk_filtered_dev(q.data_ptr(), nq, db.data_ptr(), N, L, iq.data_ptr(), idb.data_ptr(), k, idx.data_ptr(), val.data_ptr())
            assert st["filtered"], "the prefilter declined: the test would compare the exact path with itself"
        else:
            gpu_ctx.cosine_topk_dev(q.data_ptr(), nq, db.data_ptr(), N, L, iq.data_ptr(), idb.data_ptr(), k, 0, False, idx.data_ptr(), val.data_ptr())
        gpu_ctx.sync()
        res.append((idx.cpu().numpy(), val.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))


@pytest.mark.parametrize("nq,N,k", [(9, 2048, 5), (5, 30000, 10), (3, 9000, 16), (4, 2500, 1)])
def test_float64_wave_ranking_equals_the_lds_sort(gpu_ctx, nq, N, k):
    """k <= 16 over >= 2048 columns takes rank_f64_wave_kernel (one wave per row, threshold filter); PVS_OPT_TOPK_SELECT_ONLY = 1
    pins the LDS bitonic kernel.  Same lists bit for bit on duplicated rows (tied scores), a zero row, NaN / +-inf scores and an
    all-tied query (every score 0: the survivors overflow the list and the rounds fall-back ranks them)."""
    from pvsim import _ffi
    rng = np.random.default_rng(nq + N + k)
    L = 6
    db = rng.standard_normal((N, L))
    db[1::5] = db[0::5][: len(db[1::5])]
    db[7] = 0.0
    q = rng.standard_normal((nq, L))
    q[1] = 0.0                                        # all scores 0: everything ties
    q[2, 0] = np.inf                                  # non-finite scores: NaN rank last
    out = []
    for opt in (1, 0):
        gpu_ctx.set_option(_ffi.OPT_TOPK_SELECT_ONLY, opt)
        out.append(gpu_ctx.cosine_topk_f64(q, db, k))
    gpu_ctx.set_option(_ffi.OPT_TOPK_SELECT_ONLY, 0)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1].view(np.uint64), out[1][1].view(np.uint64))
    full = gpu_ctx.cosine(q[3:], db)
    order = np.argsort(-full, axis=1, kind="stable")[:, :k]
    assert np.array_equal(out[1][0][3:], order)


# ======================================================================================= Fisher: norm division inside the moments kernel
@pytest.mark.parametrize("kw", [dict(), dict(power=1.0, norm_order=1), dict(power=0.3, norm_order=3), dict(power=0.5, norm_order=np.inf)],
                         ids=["sqrt-l2", "p1-l1", "p0.3-l3", "sqrt-inf"])
def test_fisher_rows_divided_inside_the_moments_kernel_equal_the_two_pass_form(gpu_ctx, kw):
    """PVS_OPT_FISHER_SCALE = 2: one workgroup walks all dim blocks of its image and divides the row by its norm itself instead of
    the separate scale pass.  Ragged images (empty ones included; D = 96 leaves a partial dim block; K = 41 leaves rows that are
    not 16-byte aligned) through the default, through both pinned forms and in calls of 200: the rows must agree bit for bit,
    and agree with the NumPy restatement (fisher_vector.py:94-127) within the Fisher tolerance."""
    from pvsim import pack_descriptors, _ffi
    FISHER_ATOL = 1e-9
    rng = np.random.default_rng(77)
    for K, D, N in ((40, 96, 600), (41, 64, 520)):
        w = rng.random(K) + 0.2
        w /= w.sum()
        mu = rng.normal(size=(K, D))
        cov = rng.random((K, D)) + 0.3
        counts = rng.integers(0, 70, size=N)
        counts[[3, N - 1]] = 0
        imgs = [rng.normal(size=(int(c), D)).astype(np.float32) for c in counts]
        gm = gpu_ctx.gmm(w, mu, cov)
        packed, off = pack_descriptors(imgs, D, np.float32)
        one = gpu_ctx.fisher_encode(gm, packed, off, **kw)
        with gpu_ctx.option(_ffi.OPT_FISHER_SCALE, 1):
            two_pass = gpu_ctx.fisher_encode(gm, packed, off, **kw)
        with gpu_ctx.option(_ffi.OPT_FISHER_SCALE, 2):
            in_kernel = gpu_ctx.fisher_encode(gm, packed, off, **kw)
            few = gpu_ctx.fisher_encode(gm, packed[: off[5]], off[:6], **kw)
        parts = []
        for s in range(0, N, 200):
            e = min(N, s + 200)
            parts.append(gpu_ctx.fisher_encode(gm, packed[off[s]:off[e]], off[s:e + 1] - off[s], **kw))
        assert np.array_equal(one, two_pass) and np.array_equal(one, in_kernel) and np.array_equal(one, np.concatenate(parts))
        assert np.array_equal(few, one[:5])
        assert not one[3].any() and not one[N - 1].any()
        okw = dict(power=kw.get("power", 0.5), norm_order=kw.get("norm_order", 2))
        sel = [0, 1, 2, 3, 4, 300, N - 2]
        ref = orc.fisher_encode([imgs[i] for i in sel], w, mu, cov, **okw)
        np.testing.assert_allclose(one[sel], ref, rtol=0, atol=FISHER_ATOL)


# ======================================================================================= a resident index for the retrieval functions
class _LookupEncoder:
    """encode(image) -> the vector stored for that "image" (a 1 x 1 integer id array): lets the eval functions run on given vectors."""

    def __init__(self, vectors, ctx):
        self.vectors, self.context = vectors, ctx

    def encode(self, image):
        return self.vectors[int(np.asarray(image).reshape(-1)[0])].reshape(1, -1)


@pytest.mark.parametrize("dtype,nq", [(np.float32, 7), (np.float32, 640), (np.float64, 9)], ids=["f32-few", "f32-many(filtered)", "f64"])
def test_device_index_gives_the_results_of_the_dict(gpu_ctx, dtype, nq):
    """pvsim.index.DeviceIndex (the encoding map resident and normalised on the GPU) passed where eval.* take the dict
    (pyvisim/eval.py:13-145): the same paths, scores (bit for bit), mAP and accuracy as with the plain dict, in float32 (plain and
    filtered retrieval) and float64; Mapping behaviour of the index itself."""
    from pvsim import eval as pe
    from pvsim.index import DeviceIndex
    rng = np.random.default_rng(11)
    N, L = 900, 96
    db = rng.standard_normal((N, L)).astype(dtype)
    db[17] = db[3]                                              # a tie
    paths = [f"img_{i:04d}.jpg" for i in range(N)]
    labels = {p: int(i % 13) for i, p in enumerate(paths)}
    enc_map = dict(zip(paths, db))
    qv = (db[rng.integers(0, N, nq)] + 0.05 * rng.standard_normal((nq, L))).astype(dtype)
    enc = _LookupEncoder(qv, gpu_ctx)
    q_imgs = [np.array([[i]], dtype=np.int64) for i in range(nq)]
    q_lab = [int(rng.integers(0, 13)) for _ in range(nq)]
    index = DeviceIndex(enc_map, gpu_ctx)
    try:
        assert len(index) == N and list(index.keys()) == paths and np.array_equal(index[paths[5]], db[5]) and paths[7] in index
        a = pe.retrieve_top_k_similar(q_imgs[0], enc_map, enc, k=6)
        b = pe.retrieve_top_k_similar(q_imgs[0], index, enc, k=6)
        assert [p for p, _ in a] == [p for p, _ in b]
        assert np.array_equal(np.array([s for _, s in a]).view(np.uint8), np.array([s for _, s in b]).view(np.uint8))
        for k in (5, None):
            assert pe.top_k_map(q_imgs, q_lab, enc_map, labels, enc, k=k) == pe.top_k_map(q_imgs, q_lab, index, labels, enc, k=k)
        assert pe.top_k_accuracy(q_imgs, q_lab, enc_map, labels, enc, k=3) == pe.top_k_accuracy(q_imgs, q_lab, index, labels, enc, k=3)
        i1, v1 = pe._rank(qv, db, 10, gpu_ctx)
        i2, v2 = index.rank(qv, 10)
        assert np.array_equal(i1, i2) and np.array_equal(v1.view(np.uint8), v2.view(np.uint8)) and v2.dtype == dtype
    finally:
        index.close()


# ======================================================================================= one or two queries: one pass over the database
@pytest.mark.parametrize("N,L,k", [(8189, 512, 5), (300, 4096, 7), (70000, 64, 3), (33, 1032, 33), (5000, 24, 2000)])
def test_one_or_two_queries_take_the_dense_row_kernel_with_the_same_scores(gpu_ctx, N, L, k):
    """pvs_cosine_topk_dev with nq <= 2 scores the 1 x N row with the exact re-scoring kernel run densely over the whole database
    (one pass at HBM speed) instead of 128 x 128 MFMA tiles: indices and scores must be the MFMA path's bit for bit -- compared
    with the same queries ranked as rows of a 130-query call (which takes the GEMM), incl. a k-tile tail (L = 1032), several
    chains (L = 4096), a duplicated row, k = N and a deep ranking (k = 2000)."""
    rng = np.random.default_rng(N + L)
    db = rng.standard_normal((N, L)).astype(np.float32)
    db[N // 2] = db[1]
    q = (db[rng.integers(0, N, 130)] + 0.1 * rng.standard_normal((130, L))).astype(np.float32)
    q[0] = db[1]                                                # exact ties among the candidates of query 0
    d_db, d_q = gpu_ctx.buffer(db.nbytes).upload(db), gpu_ctx.buffer(q.nbytes).upload(q)
    d_invdb, d_invq = gpu_ctx.buffer(N * 4), gpu_ctx.buffer(130 * 4)
    gpu_ctx.row_inv_norms_dev(d_db.ptr, N, L, d_invdb.ptr)
    gpu_ctx.row_inv_norms_dev(d_q.ptr, 130, L, d_invq.ptr)
    d_i, d_v = gpu_ctx.buffer(130 * k * 8), gpu_ctx.buffer(130 * k * 4)
    gpu_ctx.cosine_topk_dev(d_q.ptr, 130, d_db.ptr, N, L, d_invq.ptr, d_invdb.ptr, k, 0, False, d_i.ptr, d_v.ptr)
    ri, rv = d_i.download((130, k), np.int64), d_v.download((130, k), np.float32)
    for nq in (1, 2):
        gpu_ctx.cosine_topk_dev(d_q.ptr, nq, d_db.ptr, N, L, d_invq.ptr, d_invdb.ptr, k, 0, False, d_i.ptr, d_v.ptr)
        gi, gv = d_i.download((nq, k), np.int64), d_v.download((nq, k), np.float32)
        assert np.array_equal(gi, ri[:nq]) and np.array_equal(gv.view(np.uint32), rv[:nq].view(np.uint32))
    ref = orc.cosine_similarity(q[:2], db)
    np.testing.assert_allclose(rv[:2], -np.sort(-ref, axis=1)[:, :k], rtol=0, atol=3e-6)
    for b in (d_db, d_q, d_invdb, d_invq, d_i, d_v):
        b.free()
