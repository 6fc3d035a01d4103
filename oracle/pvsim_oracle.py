"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product (pvsim/*).

A NumPy restatement of the reference's hot path.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and
only as the checker / the timed CPU baseline.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against
golden vectors produced by importing and running the reference itself in the build container
(tests/golden/make_golden.py; numpy 2.2.6, scikit-learn 1.7.2, versions in
tests/golden/versions.json).  The arithmetic that the reference delegates to scikit-learn
(un-pinned third-party dependency, setup.py:29 -- KMeans.predict, GaussianMixture.predict_proba,
PCA.transform, metrics.pairwise.cosine_similarity) is restated from the published algorithm; the
call sites are cited per function.  Paths are relative to /root/reference unless prefixed sklearn/.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "rootsift", "pca_transform", "kmeans_predict", "vlad_encode_one", "vlad_encode",
    "gmm_predict_proba", "fisher_encode_one", "fisher_encode", "cosine_similarity",
    "argsort_desc", "topk", "retrieve_top_k", "top_k_accuracy", "top_k_map", "split_ragged",
]


def split_ragged(packed: np.ndarray, offsets: np.ndarray) -> list[np.ndarray]:
    return [packed[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]


# ----------------------------------------------------------------------------- a1 RootSIFT
def rootsift(desc: np.ndarray) -> np.ndarray:
    """pyvisim/features/_features.py:112-114 -- fp32; d /= (row sum + 1e-7); sqrt."""
    d = np.array(desc, dtype=np.float32, copy=True)
    d /= (d.sum(axis=1, keepdims=True) + np.float32(1e-7))
    return np.sqrt(d)


# ----------------------------------------------------------------------------- a2 PCA.transform
def pca_transform(x: np.ndarray, components: np.ndarray, mean: np.ndarray) -> np.ndarray:
    """Call sites vlad.py:89-90, fisher_vector.py:91-92; sklearn/decomposition/_base.py:116-166:
    X @ components.T - (mean @ components.T), no whitening, fp32 in -> fp32 out."""
    x = np.asarray(x, dtype=np.float32)
    comp = np.asarray(components)
    xt = x @ comp.T
    xt -= np.reshape(mean, (1, -1)) @ comp.T
    return xt


# ----------------------------------------------------------------------------- a3 KMeans.predict
def kmeans_predict(x: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """Call site vlad.py:95; sklearn/cluster/_kmeans.py:1066-1096 -> _k_means_lloyd.pyx:168-218:
    label_i = argmin_j (||c_j||^2 - 2 x_i.c_j) in fp32, strict '<' scan => first minimum wins
    (np.argmin has the same first-occurrence rule).  ||x||^2 is not added."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    cn = np.einsum("ij,ij->i", c, c)
    pd = cn[None, :] + np.float32(-2.0) * (x @ c.T)
    return np.argmin(pd, axis=1).astype(np.int32)


def assignment_margin(x: np.ndarray, centroids: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Exact fp64 labels and best-vs-second gap: tells near-ties (where any fp32 evaluation,
    the reference's BLAS included, is free to differ) from real mismatches."""
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(centroids, dtype=np.float64)
    pd = (c * c).sum(1)[None, :] - 2.0 * (x @ c.T)
    order = np.argsort(pd, axis=1, kind="stable")[:, :2]
    best = np.take_along_axis(pd, order[:, :1], 1)[:, 0]
    second = np.take_along_axis(pd, order[:, 1:2], 1)[:, 0] if pd.shape[1] > 1 else best + np.inf
    return order[:, 0].astype(np.int32), second - best


# ----------------------------------------------------------------------------- a4+a5 VLAD
def vlad_aggregate(desc: np.ndarray, labels: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """vlad.py:98-104: V[l_i] += (x_i - c_{l_i}) sequentially in descriptor order, fp32."""
    k, d = centroids.shape
    v = np.zeros((k, d), dtype=np.float32)
    resid = desc.astype(np.float32, copy=False) - centroids[labels]     # fp32 subtract
    order = np.argsort(labels, kind="stable")                            # keeps descriptor order per cluster
    sl = labels[order]
    starts = np.searchsorted(sl, np.arange(k), side="left")
    ends = np.searchsorted(sl, np.arange(k), side="right")
    for j in np.nonzero(ends > starts)[0]:
        acc = np.zeros(d, dtype=np.float32)
        for r in resid[order[starts[j]:ends[j]]]:                        # sequential fp32 adds
            acc += r
        v[j] = acc
    return v


def vlad_normalise(v: np.ndarray, power: float, norm_order, eps: float) -> np.ndarray:
    """vlad.py:106-108: sign*|v|^p; per-cluster row norm + eps; divide.  No global L2."""
    v = np.sign(v) * np.abs(v) ** power
    norms = np.linalg.norm(v, axis=1, ord=norm_order, keepdims=True) + eps
    return v / norms


def vlad_encode_one(desc, centroids, power=1, norm_order=2, eps=1e-9, pca=None, labels_out=None):
    """vlad.py:87-111 for one image, given its (n, D) descriptors.  An empty image yields a zero
    row here (the reference aborts the whole batch with one 1-D zero vector, vlad.py:92-93 --
    quirk A.3, pinned separately in the golden `empty_quirk`)."""
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    if pca is not None:
        desc = pca_transform(np.asarray(desc, np.float32), pca[0], pca[1])
    k, d = centroids.shape
    if desc is None or len(desc) == 0:
        return np.zeros(k * d, dtype=np.float32)
    desc = np.asarray(desc, dtype=np.float32)
    labels = kmeans_predict(desc, centroids)
    if labels_out is not None:
        labels_out.append(labels)
    v = vlad_aggregate(desc, labels, centroids)
    return vlad_normalise(v, power, norm_order, eps).astype(np.float32).reshape(-1)


def vlad_encode(desc_list, centroids, power=1, norm_order=2, eps=1e-9, pca=None) -> np.ndarray:
    return np.vstack([vlad_encode_one(d, centroids, power, norm_order, eps, pca) for d in desc_list])


# ----------------------------------------------------------------------------- a6 GMM posterior
def gmm_predict_proba(x, weights, means, covariances, return_log_prob_norm=False) -> np.ndarray:
    """Call site fisher_vector.py:99.  sklearn/mixture/_base.py:393-411,513-538 and
    sklearn/mixture/_gaussian_mixture.py:413-450,495-512 ('diag'):
      prec_chol = 1/sqrt(cov); precisions = prec_chol**2; log_det = sum log prec_chol
      log_prob  = sum(mu^2 prec) - 2 X.(mu prec)^T + (X**2).prec^T     (X**2 in X's dtype!)
      logp      = -0.5 (D log(2 pi).astype(X.dtype) + log_prob) + log_det + log(weights)
      resp      = exp(logp - logsumexp_k(logp))
    fp64 tables with fp32 X give fp64 results (numpy promotion), as in the reference."""
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    means = np.asarray(means, dtype=np.float64)
    cov = np.asarray(covariances, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    n_features = means.shape[1]
    prec_chol = 1.0 / np.sqrt(cov)
    log_det = np.sum(np.log(prec_chol), axis=1)
    precisions = prec_chol ** 2
    log_prob = (np.sum(means ** 2 * precisions, 1)
                - 2.0 * np.dot(x, (means * precisions).T)
                + np.dot(x ** 2, precisions.T))
    lg = -0.5 * (n_features * np.log(2 * np.pi).astype(x.dtype) + log_prob) + log_det
    wlp = lg + np.log(weights)
    m = np.max(wlp, axis=1, keepdims=True)                       # scipy.special.logsumexp
    lse = m[:, 0] + np.log(np.sum(np.exp(wlp - m), axis=1))
    if return_log_prob_norm:
        return np.exp(wlp - lse[:, None]), lse
    return np.exp(wlp - lse[:, None])


# ----------------------------------------------------------------------------- a7+a8 Fisher
def fisher_encode_one(desc, weights, means, covariances, power=0.5, norm_order=2, eps=1e-9, pca=None):
    """fisher_vector.py:89-133 for one image.  Layout [d_pi (K) | d_mu (K*D, k-major) | d_sigma]."""
    weights = np.asarray(weights, np.float64)
    means = np.asarray(means, np.float64)
    cov = np.asarray(covariances, np.float64)
    k, d = means.shape
    if pca is not None:
        desc = pca_transform(np.asarray(desc, np.float32), pca[0], pca[1])
    n = len(desc)
    if n == 0:                      # reference: division by zero / sklearn error (A.3); defined as zero row
        return np.zeros(k + 2 * k * d, dtype=np.float64)
    g = gmm_predict_proba(desc, weights, means, cov)
    pp_sum = g.mean(axis=0, keepdims=True).T
    pp_x = g.T.dot(desc) / n
    pp_x_2 = g.T.dot(np.power(desc, 2)) / n                      # squared in desc's dtype (:104)
    d_pi = pp_sum.squeeze(axis=1) - weights
    d_mu = pp_x - pp_sum * means
    d_sigma = -pp_x_2 - pp_sum * np.power(means, 2) + pp_sum * cov + 2 * pp_x * means
    sw = np.sqrt(weights)
    d_pi = d_pi / sw
    d_mu = d_mu / (sw[:, None] * np.sqrt(cov))
    d_sigma = d_sigma / (np.sqrt(2) * sw[:, None] * cov)
    v = np.hstack((d_pi, d_mu.ravel(), d_sigma.ravel())).reshape(1, -1)
    v = np.sign(v) * np.power(np.abs(v), power)
    v = v / (np.linalg.norm(v, axis=1, ord=norm_order, keepdims=True) + eps)
    return v.reshape(-1)


def fisher_encode(desc_list, weights, means, covariances, power=0.5, norm_order=2, eps=1e-9, pca=None):
    return np.vstack([fisher_encode_one(x, weights, means, covariances, power, norm_order, eps, pca)
                      for x in desc_list])


# ----------------------------------------------------------------------------- a9 cosine
def cosine_similarity(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """pyvisim/_utils.py:312-330 -> sklearn/metrics/pairwise.py:1683-1738:
    1-D -> (1, L); ValueError if L <= 1; rows L2-normalised (zero norms -> 1,
    sklearn/preprocessing/_data.py:1912); Xn @ Yn.T; fp32 iff both operands fp32."""
    x = np.asarray(x)
    y = np.asarray(y)
    x = x.reshape(1, -1) if x.ndim == 1 else x
    y = y.reshape(1, -1) if y.ndim == 1 else y
    if x.shape[-1] <= 1 or y.shape[-1] <= 1:
        raise ValueError("Cosine similarity requires at least 2 features.")
    dt = np.float32 if (x.dtype == np.float32 and y.dtype == np.float32) else np.float64
    x = x.astype(dt, copy=False)
    y = y.astype(dt, copy=False)

    def _normalise(a):
        nrm = np.sqrt(np.einsum("ij,ij->i", a, a))
        nrm[nrm == 0.0] = 1.0
        return a / nrm[:, None]

    return _normalise(x) @ _normalise(y).T


# ----------------------------------------------------------------------------- a11 retrieval
def argsort_desc(scores: np.ndarray) -> np.ndarray:
    """eval.py:40,78,132 use np.argsort(-scores) (non-stable kind: tie order unspecified).
    The engine's defined order is (score desc, index asc) == a stable sort of -scores."""
    return np.argsort(-scores, kind="stable")


def topk(scores: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Row-wise top-k of an (nq, N) score matrix -> (indices int64 (nq,k), values)."""
    idx = np.argsort(-scores, axis=1, kind="stable")[:, :k]
    return idx.astype(np.int64), np.take_along_axis(scores, idx, axis=1)


def retrieve_top_k(query_vec, db_vecs, k=5):
    """eval.py:13-46 given the encoded query and the stacked DB vectors."""
    s = cosine_similarity(query_vec, db_vecs)[0]
    idx = argsort_desc(s)[:k]
    return idx, s[idx]


def top_k_accuracy(q_vecs, q_labels, db_vecs, db_labels, k) -> float:
    """eval.py:102-145: any label match within the first k; divides by the number of queries."""
    correct = 0
    for q, lab in zip(q_vecs, q_labels):
        s = cosine_similarity(q, db_vecs)[0]
        idx = argsort_desc(s)[:k]
        correct += int(np.any(np.asarray(db_labels)[idx] == lab))
    return correct / len(q_vecs)


def top_k_map(q_vecs, q_labels, db_vecs, db_labels, k=None) -> float:
    """eval.py:49-100: AP with R counted INSIDE the (possibly truncated) window (:95)."""
    aps = []
    db_labels = np.asarray(db_labels)
    for q, lab in zip(q_vecs, q_labels):
        s = cosine_similarity(q, db_vecs)[0]
        idx = argsort_desc(s)
        if k is not None:
            idx = idx[:k]
        rel = db_labels[idx] == lab
        cnt, psum = 0, 0.0
        for rank, r in enumerate(rel, start=1):
            if r:
                cnt += 1
                psum += cnt / rank
        big_r = int(rel.sum())
        aps.append(psum / big_r if big_r > 0 else 0.0)
    return float(np.mean(aps))


# ----------------------------------------------------------------------------- f4 learn(): the scikit-learn fits
# Call site _base_encoder.py:311-342: PCA(n_components).fit / KMeans(n_clusters, **kw).fit /
# GaussianMixture(n_components, covariance_type="diag", **kw).fit on the stacked descriptors.  The arithmetic is
# scikit-learn's (1.7.2, not vendored by the reference); restated from its published sources and pinned by the fits
# the reference's own learn() produced here (tests/golden/learn_k16_d32.npz, tests/golden/make_golden_learn.py).
def pca_fit(x: np.ndarray, n_components: int):
    """sklearn/decomposition/_pca.py:_fit_full, solver "covariance_eigh" (what "auto" selects for n >= 10 D, D <= 1000):
    everything in X's dtype; C = X^T X - n mean mean^T, / (n - 1); eigh; descending; svd_flip(u_based_decision=False)."""
    x = np.asarray(x)
    n, d = x.shape
    mean = x.mean(axis=0)
    c = x.T @ x
    c -= n * mean.reshape(-1, 1) * mean.reshape(1, -1)
    c /= n - 1
    vals, vecs = np.linalg.eigh(c)
    vals, vecs = vals[::-1].copy(), vecs[:, ::-1]
    vals[vals < 0.0] = 0.0
    vt = vecs.T.copy()
    piv = np.argmax(np.abs(vt), axis=1)
    vt *= np.sign(vt[np.arange(d), piv])[:, None]
    return vt[:n_components].copy(), mean, vals[:n_components].copy()


def kmeans_lloyd(x: np.ndarray, init: np.ndarray, max_iter: int = 300, tol: float = 1e-4, center: bool = True):
    """sklearn/cluster/_kmeans.py:KMeans.fit (init given as an array, n_init=1) -> _kmeans_single_lloyd, fp32:
    X and the start are centred by X.mean(0); tol is scaled by mean(var(X, 0)); each iteration labels with the old
    centres (a3), moves every centre to its members' mean, relocates empty clusters to the farthest points
    (_k_means_common.pyx:_relocate_empty_clusters_dense); stops when the labels repeat or sum |shift|^2 <= tol; reruns
    the labelling unless the labels repeated; inertia = sum |x - c_label|^2.  -> (centres, labels, inertia, n_iter)"""
    x = np.array(x, dtype=np.float32)
    # center=False: the same procedure in the caller's coordinates (what the device does; scikit-learn centres only to lose
    # fewer fp32 digits, and a descriptor that sits between two centres may then fall on the other side)
    mean = x.mean(axis=0) if center else np.zeros(x.shape[1], np.float32)
    x -= mean
    c = np.array(init, dtype=np.float32) - mean
    k = c.shape[0]
    tol_abs = 0 if tol == 0 else np.mean(np.var(x, axis=0)) * tol
    labels_old = np.full(x.shape[0], -1, np.int32)
    strict = False
    it = 0
    for it in range(max_iter):
        labels = kmeans_predict(x, c).astype(np.int32)
        sums = np.zeros_like(c)
        np.add.at(sums, labels, x)
        cnt = np.bincount(labels, minlength=k).astype(np.float32)
        empty = np.where(cnt == 0)[0]
        if len(empty):
            dist = ((x - c[labels]) ** 2).sum(axis=1)
            far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
            for new_id, idx in zip(empty, far):
                old = labels[idx]
                sums[old] -= x[idx]
                sums[new_id] = x[idx]
                cnt[new_id] = 1
                cnt[old] -= 1
        # _k_means_common.pyx:_average_centers: a cluster left without weight keeps its old centre
        with np.errstate(divide="ignore", invalid="ignore"):
            new = np.where(cnt[:, None] > 0, sums / cnt[:, None], c).astype(np.float32)
        shift = float(((new - c) ** 2).sum())
        c = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift <= tol_abs:
            break
        labels_old = labels
    if not strict:
        labels = kmeans_predict(x, c).astype(np.int32)
    inertia = float(((x - c[labels]).astype(np.float64) ** 2).sum())
    return c + mean, labels, inertia, it + 1


def gmm_em(x, weights, means, covariances, max_iter: int = 100, tol: float = 1e-3, reg_covar: float = 1e-6):
    """sklearn/mixture/_base.py:BaseMixture.fit_predict main loop with explicit starting tables, 'diag':
    E-step (a6) -> M-step (_gaussian_mixture.py:_estimate_gaussian_parameters: nk = sum resp + 10 eps;
    means = resp^T X / nk; cov = resp^T (X*X) / nk - means^2 + reg_covar; weights = nk / n, renormalised) ->
    lower bound = mean log p(x) under the OLD tables; stop when it changes by less than tol.
    -> (weights, means, covariances, lower_bound, n_iter, converged)"""
    x = np.asarray(x)
    n = x.shape[0]
    w, mu, cov = (np.asarray(a, dtype=np.float64) for a in (weights, means, covariances))
    lower, converged, it = -np.inf, False, 0
    for it in range(1, max_iter + 1):
        prev = lower
        resp, lpn = gmm_predict_proba(x, w, mu, cov, return_log_prob_norm=True)
        nk = resp.sum(axis=0) + 10 * np.finfo(resp.dtype).eps
        mu = np.dot(resp.T, x) / nk[:, None]
        cov = np.dot(resp.T, x * x) / nk[:, None] - mu ** 2 + reg_covar
        w = nk / n
        w /= w.sum()
        lower = float(np.mean(lpn))
        if abs(lower - prev) < tol:
            converged = True
            break
    return w, mu, cov, lower, it, converged
