"""Randomised parity sweep on the GPU (run by hand: python tests/tools/fuzz_gpu.py [seconds] [seed]).
Random shapes / kinds / normalisation parameters for VLAD, Fisher, cosine, top-k, filtered top-k and the prefiltered
assignment, each checked against the oracle or against the exact device path.  Prints the first failing case."""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()      # before the engine opens the device (the other order leaves torch without a GPU on the test box)

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle")]
import pvsim                      # noqa: E402
import pvsim_oracle as orc        # noqa: E402
import pvsim_oracle_c as orc_c    # noqa: E402
from pvsim.engine import DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = pvsim.Context(0)
t_end = time.time() + budget
counts = {}


def ragged(n_img, lo, hi):
    return [int(v) for v in rng.integers(lo, hi, n_img)]


def case_vlad():
    K = int(rng.choice([1, 7, 16, 33, 64, 100, 256, 300]))
    D = int(rng.choice([2, 8, 30, 64, 100, 128, 130, 200]))
    kind = int(rng.choice([DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT])) if D <= 128 else DESC_F32
    counts_ = ragged(int(rng.integers(1, 6)), 0, int(rng.choice([5, 300, 6000])))
    C = rng.random((K, D)).astype(np.float32) * (0.2 if kind != DESC_F32 else 1.0)
    if kind == DESC_F32:
        descs = [(C[rng.integers(0, K, n)] + 0.3 * rng.standard_normal((n, D))).astype(np.float32) for n in counts_]
        ref_in = descs
    else:
        raws = [rng.integers(0, 256, size=(n, D)).astype(np.uint8) for n in counts_]
        descs = raws if kind == DESC_U8_ROOTSIFT else [r.astype(np.float32) for r in raws]
        ref_in = [orc.rootsift(r.astype(np.float32)) for r in raws]
    power = float(rng.choice([1.0, 0.5, 0.3]))
    order = float(rng.choice([2.0, 1.0, 3.0]))
    packed, off = pvsim.pack_descriptors(descs, D, np.uint8 if kind == DESC_U8_ROOTSIFT else np.float32)
    cb = ctx.codebook(C)
    v, labels = ctx.vlad_encode(cb, packed, off, kind, power=power, norm_order=order, return_labels=True)
    cb.close()
    allx = np.concatenate(ref_in) if sum(counts_) else np.zeros((0, D), np.float32)
    want_labels = orc_c.assign_chain(allx, C) if len(allx) else np.zeros(0, np.int32)
    assert np.array_equal(labels, want_labels), ("vlad labels", K, D, kind, counts_)
    ref = np.stack([orc.vlad_normalise(orc.vlad_aggregate(x, want_labels[o:o + len(x)], C), power, order, 1e-9).reshape(-1)
                    for x, o in zip(ref_in, np.cumsum([0] + counts_[:-1]))])
    assert np.allclose(v, ref, rtol=0, atol=3e-6), ("vlad values", K, D, kind, counts_, power, order, float(np.abs(v - ref).max()))


def case_fisher():
    K = int(rng.choice([1, 5, 32, 64, 256, 300]))
    D = int(rng.choice([2, 16, 64, 100, 128, 257]))
    counts_ = ragged(int(rng.integers(1, 5)), 1, int(rng.choice([4, 200, 1500])))
    mu = rng.normal(0, 1, (K, D))
    cov = rng.uniform(0.05, 2.0, (K, D))
    w = rng.random(K) + 0.1
    w /= w.sum()
    descs = [(mu[rng.integers(0, K, n)] + rng.standard_normal((n, D)) * 0.7).astype(np.float32) for n in counts_]
    power = float(rng.choice([0.5, 1.0, 0.7]))
    order = float(rng.choice([2.0, 1.0]))
    g = ctx.gmm(w, mu, cov)
    packed, off = pvsim.pack_descriptors(descs, D, np.float32)
    scale_opt = int(rng.integers(0, 3))        # where the rows are divided by their norm: chosen / second pass / inside the kernel
    with ctx.option(pvsim._ffi.OPT_FISHER_SCALE, scale_opt):
        f = ctx.fisher_encode(g, packed, off, power=power, norm_order=order)
    if scale_opt:
        assert np.array_equal(f, ctx.fisher_encode(g, packed, off, power=power, norm_order=order)), ("fisher scale forms differ", K, D, counts_, scale_opt)
    g.close()
    ref = orc.fisher_encode(descs, w, mu, cov, power=power, norm_order=order)
    assert np.allclose(f, ref, rtol=0, atol=2e-9), ("fisher", K, D, counts_, power, order, float(np.abs(f - ref).max()))


def case_cosine_topk():
    M, N = int(rng.integers(1, 400)), int(rng.integers(1, 3000))
    L = int(rng.choice([2, 24, 40, 128, 1000, 1024, 2600, 4096, 32768]))
    same = bool(rng.integers(0, 2))
    a = rng.standard_normal((M, L)).astype(np.float32)
    b = a if same else rng.standard_normal((N, L)).astype(np.float32)
    if same:
        N = M
    s = ctx.cosine(a, b)
    ref = orc.cosine_similarity(a, b)
    # NumPy normalises the rows first and then multiplies: its own self-scores reach 1.000003 at L = 32768
    assert np.allclose(s, ref, rtol=0, atol=5e-6 if L >= 32768 else 3e-6), ("cosine", M, N, L, same, float(np.abs(s - ref).max()))
    k = int(rng.integers(1, min(N, 40) + 1))
    idx, val = ctx.cosine_topk(a, b, k)
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    assert np.array_equal(idx, order), ("topk", M, N, L, k)


def case_cosine_f64():
    """float64 operands: the f64 MFMA GEMM (even L, general / symmetric / split-K) or the vector-ALU kernel (odd L) + float64 ranking"""
    M, N = int(rng.integers(1, 600)), int(rng.integers(1, 5000))
    L = int(rng.choice([2, 3, 24, 130, 1000, 1023, 4098, 33000]))
    if L >= 33000:
        M, N = min(M, 200), min(N, 1500)
    same = bool(rng.integers(0, 2))
    a = rng.standard_normal((M, L)) * float(rng.choice([1e-6, 1.0, 1e5]))
    b = a if same else rng.standard_normal((N, L))
    if same:
        N = M
    if N > 3 and rng.integers(0, 2):
        b[1] = b[0]
        b[2] = 0.0
    s = ctx.cosine(a, b)
    ref = orc.cosine_similarity(a, b)
    assert s.dtype == np.float64 and np.allclose(s, ref, rtol=0, atol=1e-13), ("cosine f64", M, N, L, same, float(np.abs(s - ref).max()))
    if same:
        assert np.array_equal(s, s.T), ("cosine f64 symmetry", M, L)
    k = N if rng.integers(0, 4) == 0 else int(rng.integers(1, min(N, 60) + 1))
    idx, val = ctx.cosine_topk_f64(a, b, k)
    # (the host top-k form computes its own panels: general kernel for query blocks, symmetric when Q is DB -- same bits as `s`
    # only where the same kernel and split produced them, so rank the scores it returns itself against `s` by value)
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    want = np.take_along_axis(s, order, 1)
    assert np.allclose(val, want, rtol=0, atol=1e-13), ("topk f64 values", M, N, L, k)
    clear = np.ones_like(order, dtype=bool)
    if k > 1:
        gap = np.abs(np.diff(want, axis=1)) > 1e-12
        clear[:, 1:] &= gap
        clear[:, :-1] &= gap
    if k < N:       # the first excluded score must be clearly below the last included one for that rank to be pinned
        nxt = np.take_along_axis(s, np.argsort(-s, axis=1, kind="stable")[:, k:k + 1], 1)
        clear[:, -1:] &= (want[:, -1:] - nxt) > 1e-12
    assert np.array_equal(idx[clear], order[clear]), ("topk f64 order", M, N, L, k)


def case_filtered():
    nq, N = int(rng.integers(1, 700)), int(rng.integers(2, 4000))
    L = int(rng.choice([8, 64, 1000, 1024, 2600, 4096, 8192]))
    if rng.integers(0, 12) == 0:                # now and then a database of several score panels (running threshold)
        N, L = int(rng.integers(33000, 90000)), int(rng.choice([8, 64, 256]))
    same = bool(rng.integers(0, 2))
    base = rng.standard_normal((N, L)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 1e3]))
    if rng.integers(0, 2):
        base[rng.integers(0, N, N // 4)] = base[rng.integers(0, N, N // 4)]      # duplicate rows
    db = torch.from_numpy(base).cuda()
    q = db if same else torch.from_numpy((base[rng.integers(0, N, nq)] * (1 + 1e-3 * rng.standard_normal((nq, L)))).astype(np.float32)).cuda()
    nq = q.shape[0]
    k = int(rng.integers(1, min(N, 30) + 1))
    iq = torch.empty(nq, dtype=torch.float32, device="cuda")
    idb = iq if same else torch.empty(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.row_inv_norms_dev(q.data_ptr(), nq, L, iq.data_ptr())
    if not same:
        ctx.row_inv_norms_dev(db.data_ptr(), N, L, idb.data_ptr())
    res = []
    for filt in (0, 1):
        idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        val = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if filt:
            ctx.cosine_topk_filtered_dev(q.data_ptr(), nq, db.data_ptr(), N, L, iq.data_ptr(), idb.data_ptr(), k, idx.data_ptr(), val.data_ptr())
        else:
            ctx.cosine_topk_dev(q.data_ptr(), nq, db.data_ptr(), N, L, iq.data_ptr(), idb.data_ptr(), k, 0, False, idx.data_ptr(), val.data_ptr())
        ctx.sync()
        res.append((idx.cpu().numpy(), val.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]), ("filtered idx", nq, N, L, k, same)
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32)), ("filtered val", nq, N, L, k, same)


def case_learn():
    from pvsim import learn
    import warnings
    K = int(rng.choice([2, 5, 16, 40, 64]))
    D = int(rng.choice([2, 8, 30, 64, 128]))
    n = int(rng.choice([300, 5000, 20000]))
    mu = rng.normal(0, 3, (K, D))
    x = (mu[rng.integers(0, K, n)] + rng.standard_normal((n, D))).astype(np.float32)
    rows = learn.DeviceRows.from_host(ctx, x)
    c0 = x[rng.choice(n, K, replace=False)].copy()
    m = learn.fit_kmeans(rows, K, init=c0, n_init=1, max_iter=3, tol=0.0)
    rc, rl, _, _ = orc.kmeans_lloyd(x, c0, max_iter=3, tol=0.0, center=False)
    if np.mean(m.labels_ != rl) >= 5e-3:
        bad = np.where(m.labels_ != rl)[0]
        _, gap = orc.assignment_margin(x, rc)
        m1 = learn.fit_kmeans(rows, K, init=c0, n_init=1, max_iter=1, tol=0.0)
        r1 = orc.kmeans_lloyd(x, c0, max_iter=1, tol=0.0, center=False)
        lab_b = ctx.buffer(n * 4)
        cbk = ctx.codebook(c0)
        resid, counts, inertia, _ = ctx.kmeans_step_dev(cbk, rows.ptr, n, lab_b.ptr, None, None)
        l0 = lab_b.download((n,), np.int32)
        ref_l0 = orc.kmeans_predict(x, c0)
        xc = x - x.mean(axis=0)
        ref_l0c = orc.kmeans_predict(xc, c0 - x.mean(axis=0))
        for it_ in (2, 3):
            mi = learn.fit_kmeans(rows, K, init=c0, n_init=1, max_iter=it_, tol=0.0)
            ri = orc.kmeans_lloyd(x, c0, max_iter=it_, tol=0.0, center=False)
            print("DIAG iters", it_, "mismatches", int((mi.labels_ != ri[1]).sum()), "n_iter device/restatement", mi.n_iter_, ri[3],
                  "nan in device centres", bool(np.isnan(mi.cluster_centers_).any()), "nan in restatement centres", bool(np.isnan(ri[0]).any()),
                  "empty clusters device/restatement", int((np.bincount(mi.labels_, minlength=K) == 0).sum()),
                  int((np.bincount(ri[1], minlength=K) == 0).sum()), flush=True)
        b0 = np.where(l0 != ref_l0)[0]
        if len(b0):
            d64 = ((x[b0].astype(np.float64)[:, None, :] - c0.astype(np.float64)[None, :, :]) ** 2).sum(-1)
            ds = np.sort(d64, axis=1)
            print("DIAG step0 mismatching points", b0[:8], "fp64 gap between their two nearest centres", (ds[:, 1] - ds[:, 0])[:8],
                  "scale |x||c|", (np.linalg.norm(x[b0], axis=1) * np.linalg.norm(c0, axis=1).max())[:8],
                  "device label / restatement label / fp64 argmin", l0[b0][:8], ref_l0[b0][:8], d64.argmin(1)[:8], flush=True)
        print("DIAG step0: device labels == plain restatement", np.array_equal(l0, ref_l0), "== centred restatement", np.array_equal(l0, ref_l0c),
              "counts", counts, "restatement counts (centred)", np.bincount(ref_l0c, minlength=K), flush=True)
        print("DIAG labels: mismatches", len(bad), "margins of the mismatched points under the restatement's centres", gap[bad][:8],
              "after ONE iteration: label mismatches", int(np.sum(m1.labels_ != r1[1])), "centre diff", float(np.abs(m1.cluster_centers_ - r1[0]).max()), flush=True)
    if np.mean(m.labels_ != rl) >= 5e-3:
        # Lloyd's iteration amplifies a single flipped point when two centres share a true cluster: accept the case when every
        # first-step difference is a near tie that no fp32 evaluation resolves (fp64 gap of the two nearest centres < 2e-6 |x||c|)
        b0 = np.where(l0 != ref_l0)[0]
        d64 = ((x[b0].astype(np.float64)[:, None, :] - c0.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        ds = np.sort(d64, axis=1)
        scale = np.linalg.norm(x[b0], axis=1) * np.linalg.norm(c0, axis=1).max()
        if len(b0) and np.all(ds[:, 1] - ds[:, 0] < 2e-6 * scale):
            counts["near_tie_amplified"] = counts.get("near_tie_amplified", 0) + 1
            rows.free()
            return
    assert np.mean(m.labels_ != rl) < 5e-3, ("kmeans labels", K, D, n, float(np.mean(m.labels_ != rl)))
    if np.array_equal(m.labels_, rl):
        # equal FINAL labels do not exclude one near-tie descriptor on the other side in an intermediate E-step: that moves a
        # centre by (its distance) / (member count)
        tol_c = (1 + np.abs(rc).max()) * (2e-4 + 3.0 / max(1, np.bincount(rl, minlength=K).min()))
        ok = np.allclose(m.cluster_centers_, rc, rtol=0, atol=tol_c)
        if not ok:
            bad = np.where(np.abs(m.cluster_centers_ - rc).max(1) > tol_c)[0]
            cnt = np.bincount(rl, minlength=K)
            print("DIAG centres differ in clusters", bad, "member counts", cnt[bad], "device", m.cluster_centers_[bad][:2], "restatement", rc[bad][:2], flush=True)
        assert ok, ("kmeans centres", K, D, n)
    Kg = min(K, 16)
    w0 = np.full(Kg, 1.0 / Kg)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w, mu_, cov, lower, _, _ = orc.gmm_em(x, w0, c0[:Kg].astype(np.float64), np.ones((Kg, D)), max_iter=3, tol=0.0)
        try:
            g = learn.fit_gmm(rows, Kg, weights_init=w0, means_init=c0[:Kg].astype(np.float64), precisions_init=np.ones((Kg, D)),
                              max_iter=3, tol=0.0)
        except ValueError:
            # scikit-learn raises the same error when a covariance is not positive: the restatement must show it too
            assert not np.all(np.isfinite(cov)) or np.any(cov <= 0), ("gmm raised but the restatement is fine", Kg, D, n)
            rows.free()
            return
    assert np.allclose(g.means_, mu_, rtol=1e-8, atol=1e-10) and np.allclose(g.covariances_, cov, rtol=1e-8, atol=1e-10), ("gmm", Kg, D, n)
    assert abs(g.lower_bound_ - lower) < 1e-8 * (1 + abs(lower)), ("gmm lower bound", Kg, D, n, g.lower_bound_, lower)
    if n >= 10 * D and D >= 2:
        p = learn.fit_pca(rows, max(1, D // 2))
        c64, _, ev = orc.pca_fit(x.astype(np.float64), max(1, D // 2))
        assert np.allclose(p.explained_variance_, ev, rtol=1e-8), ("pca variance", D, n)
    rows.free()


def case_single_query():
    """One or two queries (the dense one-pass kernel where the shape qualifies) and a resident index (pvsim.index.DeviceIndex) against
    the same queries ranked as rows of a larger call (the MFMA tiles) / against the plain matrix: indices and scores bit for bit."""
    from pvsim.index import DeviceIndex
    N = int(rng.choice([1, 7, 130, 2000, 9000, 40000]))
    L = int(rng.choice([8, 24, 30, 40, 100, 128, 1000, 1032, 4096, 32768]))      # 30 and 100: shapes the dense kernel declines
    if N * L > 60_000_000:
        L = 128
    db = rng.standard_normal((N, L)).astype(np.float32)
    if N > 3:
        db[N // 2] = db[1]
    q = (db[rng.integers(0, N, 140)] + 0.05 * rng.standard_normal((140, L))).astype(np.float32)
    k = int(rng.integers(1, min(N, 50) + 1))
    ri, rv = ctx.cosine_topk(q, db, k)                       # 140 queries: the GEMM (or filtered) path
    for nq in (1, 2):
        gi, gv = ctx.cosine_topk(q[:nq], db, k)
        assert np.array_equal(gi, ri[:nq]) and np.array_equal(gv.view(np.uint32), rv[:nq].view(np.uint32)), ("single query", N, L, k, nq)
    index = DeviceIndex({f"p{i}": db[i] for i in range(N)}, ctx)
    try:
        nq = int(rng.choice([1, 2, 5, 140]))
        gi, gv = index.rank(q[:nq], k)
        assert np.array_equal(gi, ri[:nq]) and np.array_equal(gv.view(np.uint32), rv[:nq].view(np.uint32)), ("resident index", N, L, k, nq)
    finally:
        index.close()


cases = [case_vlad, case_fisher, case_cosine_topk, case_cosine_f64, case_filtered, case_learn, case_single_query]
if os.environ.get("FUZZ_ONLY"):
    cases = [c for c in cases if c.__name__ == os.environ["FUZZ_ONLY"]]
i = 0
t_say = time.time() + 30.0
while time.time() < t_end:
    if time.time() > t_say:          # a line every half minute: a silent long run looks hung to a job runner
        print("fuzz progress", counts, flush=True)
        t_say = time.time() + 30.0
    fn = cases[i % len(cases)]
    i += 1
    try:
        fn()
    except AssertionError as e:
        print("FAIL", fn.__name__, "seed", seed, "iteration", i, e.args[0] if e.args else "")
        sys.exit(1)
    counts[fn.__name__] = counts.get(fn.__name__, 0) + 1
print("fuzz ok", counts)
