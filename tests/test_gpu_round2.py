"""GPU parity tests added in round 2 (through the C-ABI): the BASELINE configs[2] shape, eval.* with Fisher vectors in
float64, the float64 ranking kernel, the stand-alone `_dev` entry points, the strict-compat quirk, two contexts in one process."""
import numpy as np
import pytest

import pvsim_oracle as orc
from conftest import load_golden
from pvsim import synth, pack_descriptors, _ffi
from pvsim.engine import DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT

pytestmark = pytest.mark.gpu
FISHER_ATOL = 1e-9


def _raws(g, key_raw="raw_u8", key_off="offsets"):
    return orc.split_ragged(g[key_raw], g[key_off])


# ======================================================================================= BASELINE configs[2]: Fisher D=512 K=256 n=196
def test_fisher_config3_shape_against_the_reference_and_the_oracle(gpu_ctx):
    """12 images x 196 descriptors x 512 dims, diagonal GMM K = 256 (seeded synthetic tables): the HIP Fisher path against the
    reference's own output (fixture: image 0 whole, every 32nd element of all images, 12 x 12 float64 cosine) and against the
    NumPy restatement on every element; float64, 1e-9 abs."""
    from config3_inputs import config3_inputs
    g = load_golden("fisher_k256_d512")
    w, mu, cov, imgs = config3_inputs(12)
    gm = gpu_ctx.gmm(w, mu, cov)
    packed, off = pack_descriptors(imgs, 512, np.float32)
    f = gpu_ctx.fisher_encode(gm, packed, off)
    assert f.dtype == np.float64 and f.shape == (12, 256 + 2 * 256 * 512)
    np.testing.assert_allclose(f[0], g["image0"], rtol=0, atol=FISHER_ATOL)
    np.testing.assert_allclose(f[:, ::32], g["every32"], rtol=0, atol=FISHER_ATOL)
    np.testing.assert_allclose(f, orc.fisher_encode(imgs, w, mu, cov), rtol=0, atol=FISHER_ATOL)
    np.testing.assert_allclose(gpu_ctx.cosine(f, f), g["cos_self"], rtol=0, atol=1e-9)      # float64 operands -> float64 scores
    idx, val = gpu_ctx.cosine_topk_f64(f, f, 3)
    assert val.dtype == np.float64 and np.array_equal(idx[:, 0], np.arange(12))
    ref = np.argsort(-g["cos_self"], axis=1, kind="stable")[:, :3]
    assert np.array_equal(idx, ref)


def test_fisher_config3_full_size_properties(gpu_ctx):
    """8189 images at the configs[2] shape on the device (fp32 stored encodings, as bench.py --workload fisher keeps them):
    unit L2 norm of every row (the Fisher normalisation is global), finite values, self-retrieval; a 16-image subsample equals
    the restatement to 1e-6 relative to the unit norm (fp32 storage)."""
    import torch
    from config3_inputs import config3_inputs
    w, mu, cov, _ = config3_inputs(0)
    N, n, D, K = 8189, 196, 512, 256
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    gm = gpu_ctx.gmm(w, mu, cov)
    z = torch.randint(0, K, (N * n,), generator=g, device=dev)
    desc = (torch.from_numpy(mu.astype(np.float32)).to(dev)[z] +
            torch.from_numpy(np.sqrt(cov).astype(np.float32)).to(dev)[z] * torch.randn((N * n, D), generator=g, device=dev)).contiguous()
    off = (torch.arange(0, N + 1, dtype=torch.int64, device=dev) * n).contiguous()
    L = K + 2 * K * D
    enc = torch.empty((N, L), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.fisher_encode_dev(gm, desc.data_ptr(), DESC_F32, off.data_ptr(), N, N * n, enc.data_ptr(), 0)
    gpu_ctx.sync()
    assert bool(torch.isfinite(enc).all())
    nrm = enc.double().norm(dim=1)
    assert float((nrm - 1.0).abs().max()) < 1e-5
    sub = [desc[i * n:(i + 1) * n].cpu().numpy() for i in range(0, N, 512)]
    ref = orc.fisher_encode(sub, w, mu, cov)
    got = enc[::512].cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    inv = torch.empty((N,), dtype=torch.float32, device=dev)
    idx = torch.empty((256, 5), dtype=torch.int64, device=dev)
    val = torch.empty((256, 5), dtype=torch.float32, device=dev)
    gpu_ctx.row_inv_norms_dev(enc.data_ptr(), N, L, inv.data_ptr())
    gpu_ctx.cosine_topk_dev(enc.data_ptr(), 256, enc.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), 5, 0, False, idx.data_ptr(), val.data_ptr())
    gpu_ctx.sync()
    assert np.array_equal(idx[:, 0].cpu().numpy(), np.arange(256))


# ======================================================================================= eval.* with Fisher vectors: float64
def test_api_eval_functions_with_fisher_vectors(tables):
    """pyvisim's eval functions driven with a FisherVectorEncoder: encodings are float64, so the reference scores AND ranks in
    float64 (pyvisim/_utils.py:312-330, eval.py:37-43); lists bit-identical, scores float64 within 1e-9, acc / mAP equal."""
    from pvsim import eval as ev
    from pvsim.encoders import FisherVectorEncoder
    from pvsim.features import Lambda
    from pvsim.models import GMMModel
    g, e = load_golden("eval_fisher_db64"), load_golden("eval_db64")
    fenc = FisherVectorEncoder(Lambda(synth.rootsift, 128),
                               gmm_model=GMMModel(tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"]))
    db = [r.astype(np.float32) for r in _raws(e, "db_raw_u8", "db_offsets")]
    qs = [r.astype(np.float32) for r in _raws(e, "q_raw_u8", "q_offsets")]
    paths = [f"img_{i:03d}.jpg" for i in range(len(db))]
    fdb = fenc.encode(db)
    assert fdb.dtype == np.float64
    emap = dict(zip(paths, fdb))
    plab = dict(zip(paths, [int(l) for l in e["db_labels"]]))
    for qi in range(len(qs)):
        top = ev.retrieve_top_k_similar([qs[qi]], emap, fenc, k=7)
        assert [paths.index(p) for p, _ in top] == list(g["top7_index"][qi])
        assert all(isinstance(s, np.float64) for _, s in top)
        np.testing.assert_allclose([s for _, s in top], g["top7_score"][qi], rtol=0, atol=1e-9)
    wrapped = [[q] for q in qs]
    labels = list(e["q_labels"])
    assert ev.top_k_accuracy(wrapped, labels, emap, plab, fenc, 1) == float(g["acc_k1"])
    assert ev.top_k_accuracy(wrapped, labels, emap, plab, fenc, 5) == float(g["acc_k5"])
    assert abs(ev.top_k_map(wrapped, labels, emap, plab, fenc, 5) - float(g["map_k5"])) < 1e-12
    assert abs(ev.top_k_map(wrapped, labels, emap, plab, fenc, None) - float(g["map_all"])) < 1e-12


@pytest.mark.parametrize("nq,N,k", [(5, 10, 3), (3, 8189, 8189), (2, 20000, 4096), (4, 300, 300), (6, 9000, 1), (2, 8192, 100),
                                    (2, 9000, 9000), (3, 20000, 5000), (1, 8193, 8193)])
def test_float64_ranking_equals_a_stable_argsort(gpu_ctx, nq, N, k):
    """pvs_cosine_topk_f64: (score descending, index ascending) with NaN last, on rows with exact ties, -0 / +0, +-inf;
    several LDS chunks when N > 8192; rankings deeper than 4096 over more than 8192 columns (top_k_map(k=None) on a Fisher
    database: a full argsort in the reference, eval.py:78) page through the complete rows."""
    rng = np.random.default_rng(nq * 31 + N + k)
    L = 6
    db = rng.standard_normal((N, L))
    db[1::7] = db[0::7][: len(db[1::7])]            # duplicated rows -> exactly tied scores
    if N > 6:
        db[5] = 0.0                                  # zero row: score 0 against everything
    q = rng.standard_normal((nq, L))
    idx, val = gpu_ctx.cosine_topk_f64(q, db, k)
    s = orc.cosine_similarity(q, db)
    assert s.dtype == np.float64
    # the device's own fp64 scores may differ from NumPy's in the last bits: rank the device scores (stable) and compare
    full = gpu_ctx.cosine(q, db)
    np.testing.assert_allclose(full, s, rtol=0, atol=1e-13)
    order = np.argsort(-full, axis=1, kind="stable")[:, :k]
    assert np.array_equal(idx, order)
    assert np.array_equal(val, np.take_along_axis(full, order, 1))


# ======================================================================================= stand-alone _dev entry points
def test_kmeans_predict_dev_alone(gpu_ctx, tables):
    """pvs_kmeans_predict_dev (KMeans.predict alone, vlad.py:95) against the reference's labels of the golden images."""
    import torch
    g = load_golden("vlad_k256_d128")
    dev = torch.device("cuda", 0)
    cb = gpu_ctx.codebook(tables["centroids"])
    for kind, arr in ((DESC_U8_ROOTSIFT, g["raw_u8"]), (DESC_F32_ROOTSIFT, g["raw_u8"].astype(np.float32)),
                      (DESC_F32, orc.rootsift(g["raw_u8"].astype(np.float32)))):
        d_x = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        lab = torch.full((len(arr),), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        gpu_ctx.kmeans_predict_dev(cb, d_x.data_ptr(), kind, len(arr), lab.data_ptr())
        gpu_ctx.sync()
        assert np.array_equal(lab.cpu().numpy(), g["labels"])


def test_gmm_predict_proba_dev_alone(gpu_ctx, tables):
    """pvs_gmm_predict_proba_dev against the reference's GaussianMixture.predict_proba output (fixture resp_img3)."""
    import torch
    g, v = load_golden("fisher_k256_d128"), load_golden("vlad_k256_d128")
    x = orc.rootsift(_raws(v)[3].astype(np.float32))
    dev = torch.device("cuda", 0)
    gm = gpu_ctx.gmm(tables["gmm_weights"], tables["gmm_means"], tables["gmm_covariances"])
    d_x = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    resp = torch.empty((len(x), 256), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.gmm_predict_proba_dev(gm, d_x.data_ptr(), DESC_F32, len(x), resp.data_ptr())
    gpu_ctx.sync()
    r = resp.cpu().numpy()
    np.testing.assert_allclose(r, g["resp_img3"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r.sum(axis=1), 1.0, rtol=0, atol=1e-12)


def test_pca_transform_dev_alone(gpu_ctx, tables):
    """pvs_pca_transform_dev against PCA.transform as restated by the oracle (pinned on the reference's PCA goldens)."""
    import torch
    v = load_golden("vlad_k256_d128")
    dev = torch.device("cuda", 0)
    raw = v["raw_u8"][:3000]
    p = gpu_ctx.pca(tables["pca_components"], tables["pca_mean"])
    for kind, arr in ((DESC_U8_ROOTSIFT, raw), (DESC_F32, orc.rootsift(raw.astype(np.float32)))):
        d_x = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        out = torch.empty((len(arr), 64), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        gpu_ctx.pca_transform_dev(p, d_x.data_ptr(), kind, len(arr), out.data_ptr())
        gpu_ctx.sync()
        ref = orc.pca_transform(orc.rootsift(raw.astype(np.float32)), tables["pca_components"], tables["pca_mean"])
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=5e-6)


def test_cosine_of_two_1d_vectors_and_strict_compat(tables):
    """cos_1d: cosine_similarity of two 1-D vectors is (1, 1) (pyvisim/_utils.py:312-330); strict_compat=True reproduces the
    reference's empty-image quirk (vlad.py:92-93: the whole batch collapses to ONE 1-D zero vector) on the HIP path."""
    from pvsim._utils import cosine_similarity
    from pvsim.encoders import VLADEncoder
    from pvsim.features import Lambda
    from pvsim.models import KMeansModel
    c = load_golden("cosine")
    s = cosine_similarity(c["a32"][0], c["b32"][1])
    assert s.shape == c["cos_1d"].shape and s.dtype == c["cos_1d"].dtype
    np.testing.assert_allclose(s, c["cos_1d"], rtol=0, atol=2e-6)
    g = load_golden("small_k16_d8")
    raws = orc.split_ragged(g["raw"], g["offsets"])
    ident = Lambda(lambda im: im.astype(np.float32) / np.float32(16.0), 8)
    batch = [raws[0], np.zeros((0, 8), np.float32), raws[1]]
    strict = VLADEncoder(ident, kmeans_model=KMeansModel(g["centroids"]), strict_compat=True).encode(batch)
    assert strict.shape == g["empty_quirk"].shape and np.array_equal(strict, g["empty_quirk"])
    loose = VLADEncoder(ident, kmeans_model=KMeansModel(g["centroids"])).encode(batch)
    assert loose.shape == (3, 16 * 8) and not loose[1].any()
    np.testing.assert_allclose(loose[[0, 2]], g["vlad_default"][:2], rtol=0, atol=5e-7)


# ======================================================================================= several contexts in one process
def test_two_contexts_interleaved_on_their_own_streams(tables):
    """Two contexts on one device, each with its own stream, issue similarity GEMMs of different shapes alternately (the tile
    lists and the dynamic-LDS limits are per context): every result equals the single-context result bit for bit."""
    import torch
    import pvsim
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    shapes = [(300, 700, 512), (129, 129, 4096), (1000, 64, 256), (257, 300, 1024)]
    data = [(torch.from_numpy(rng.standard_normal((m, l)).astype(np.float32)).to(dev),
             torch.from_numpy(rng.standard_normal((n, l)).astype(np.float32)).to(dev)) for m, n, l in shapes]
    torch.cuda.synchronize()

    def run(ctx, a, b):
        m, n, l = a.shape[0], b.shape[0], a.shape[1]
        ia = torch.empty((m,), dtype=torch.float32, device=dev)
        ib = torch.empty((n,), dtype=torch.float32, device=dev)
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.row_inv_norms_dev(a.data_ptr(), m, l, ia.data_ptr())
        ctx.row_inv_norms_dev(b.data_ptr(), n, l, ib.data_ptr())
        ctx.cosine_dev(a.data_ptr(), m, b.data_ptr(), n, l, ia.data_ptr(), ib.data_ptr(), out.data_ptr(), n)
        return out, (ia, ib)

    with pvsim.Context(0) as solo:
        ref = []
        for a, b in data:
            o, keep = run(solo, a, b)
            solo.sync()
            ref.append(o.cpu().numpy())
    c1, c2 = pvsim.Context(0), pvsim.Context(0)
    assert c1.stream != c2.stream
    outs, keeps = [], []
    for rep in range(3):
        for i, (a, b) in enumerate(data):
            o, keep = run(c1 if (i + rep) % 2 == 0 else c2, a, b)       # no synchronisation between the launches
            outs.append((i, o))
            keeps.append(keep)
    c1.sync(); c2.sync()
    for i, o in outs:
        assert np.array_equal(o.cpu().numpy().view(np.uint32), ref[i].view(np.uint32))
    c1.close(); c2.close()


@pytest.mark.parametrize("order", ["engine_first", "torch_first"])
def test_engine_and_torch_share_one_hip_runtime(order):
    """Either import / initialisation order: one HIP runtime image mapped, torch sees the GPU, and the engine works on a
    torch tensor's device pointer (row norms of a CUDA tensor) in the same process."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    pkg = os.path.join(REPO, "python-visual-similarity_amd")
    eng = "import pvsim; ctx = pvsim.Context(0)"
    tor = "import torch; assert torch.cuda.is_available(); x = torch.arange(12, dtype=torch.float32, device='cuda').reshape(3, 4).contiguous()"
    first, second = (eng, tor) if order == "engine_first" else (tor, eng)
    code = (f"import sys; sys.path.insert(0, {pkg!r}); {first}; {second}; from pvsim import _ffi; "
            "inv = torch.empty(3, dtype=torch.float32, device='cuda'); torch.cuda.synchronize(); "
            "ctx.row_inv_norms_dev(x.data_ptr(), 3, 4, inv.data_ptr()); ctx.sync(); "
            "ref = 1.0 / x.norm(dim=1); assert torch.allclose(inv, ref, rtol=1e-6), (inv, ref); "
            "print('RUNTIMES', len(_ffi.mapped_hip_runtimes()))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert [l for l in r.stdout.splitlines() if l.startswith("RUNTIMES")][-1].split()[1] == "1", r.stdout


# ======================================================================================= deep features stay on the device
def test_deep_features_reach_the_encoder_without_a_host_round_trip():
    """FisherVectorEncoder / VLADEncoder .encode with a DeepConvFeature extractor: the (B, 196, D) feature tensor of
    extractor.batch() is handed to pvs_fisher_encode_dev / pvs_vlad_encode_dev as a device pointer.  Equal (1e-12 / bit for
    bit) to feeding the SAME features through the host entry point; against the per-image path (the reference's call
    sequence) only up to the convolution's batch-size dependent rounding.  Extractor parity with torchvision's VGG16 is
    unpinned (torchvision and its weights are absent offline; own `features` stack, random weights)."""
    import torch
    from pvsim.features import DeepConvFeature
    from pvsim.encoders import FisherVectorEncoder, VLADEncoder
    from pvsim.models import GMMModel, KMeansModel
    torch.manual_seed(0)
    fx = DeepConvFeature(spatial_encoding=False, device="cuda")
    assert fx.output_dim == 512
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, size=(80 + 8 * i, 100, 3), dtype=np.uint8) for i in range(5)]
    K, D = 16, 512
    gm = GMMModel(np.full(K, 1.0 / K), rng.normal(0, 0.05, (K, D)), rng.uniform(0.01, 0.05, (K, D)))
    fenc = FisherVectorEncoder(fx, gmm_model=gm)
    calls = []
    orig = fx.__class__.__call__

    def counting(self, image):
        calls.append(1)
        return orig(self, image)

    fx.__class__.__call__ = counting
    try:
        f_dev = fenc.encode(imgs)
    finally:
        fx.__class__.__call__ = orig
    assert not calls, "the per-image host path ran"
    feats = fx.batch(imgs).cpu().numpy()
    f_host = fenc.encode_descriptors([feats[i] for i in range(len(imgs))])
    assert f_dev.dtype == np.float64 and f_dev.shape == (5, K + 2 * K * D)
    np.testing.assert_allclose(f_dev, f_host, rtol=0, atol=1e-12)
    f_each = np.vstack([fenc.encode_descriptors([fx(im)]) for im in imgs])
    np.testing.assert_allclose(f_dev, f_each, rtol=0, atol=1e-5)
    venc = VLADEncoder(fx, kmeans_model=KMeansModel(rng.normal(0, 0.05, (K, D)).astype(np.float32)))
    v_dev = venc.encode(imgs)
    v_host = venc.encode_descriptors([feats[i] for i in range(len(imgs))])
    assert np.array_equal(v_dev, v_host)


def test_sharded_index_without_torch(gpu_ctx, tables):
    """pvsim.distributed.ShardedVLADIndex on one GPU with no torch object anywhere: pvs_malloc memory (DevArray / DevicePool),
    descriptors uploaded through the C-ABI, encode -> (no exchange at world 1) -> all-vs-all top-k; equal to the host entry points."""
    from pvsim import distributed as pd
    rng = np.random.default_rng(12)
    imgs = [synth.sift_like(int(n), rng).astype(np.uint8) for n in rng.integers(40, 400, size=37)]
    packed, off = pack_descriptors(imgs, 128, np.uint8)
    cb = gpu_ctx.codebook(tables["centroids"])
    pool = pd.DevicePool(gpu_ctx)
    d_x = pool.empty(packed.shape, "uint8").upload(packed)
    d_off = pool.empty(off.shape, "int64").upload(off)
    index = pd.ShardedVLADIndex(gpu_ctx, cb, len(imgs))
    index.encode_local(d_x.ptr, DESC_U8_ROOTSIFT, d_off.ptr, int(off[-1]))
    index.exchange()
    idx, val = index.topk(4)
    enc = gpu_ctx.vlad_encode(cb, packed, off, DESC_U8_ROOTSIFT)
    ridx, rval = gpu_ctx.cosine_topk(enc, enc, 4)
    assert np.array_equal(index.enc_loc.numpy()[: len(imgs)], enc)
    assert np.array_equal(idx, ridx) and np.array_equal(val.view(np.uint32), rval.view(np.uint32))
    pool.close()
