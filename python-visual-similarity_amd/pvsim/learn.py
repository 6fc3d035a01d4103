"""Vocabulary training on the device: PCA, k-means (Lloyd + k-means++) and diagonal-covariance GMM (EM).

The reference's ``ImageEncoderBase.learn`` (pyvisim/encoders/_base_encoder.py:311-342) stacks the descriptors of all
training images and calls ``sklearn.decomposition.PCA.fit`` / ``sklearn.cluster.KMeans.fit`` /
``sklearn.mixture.GaussianMixture(covariance_type="diag").fit`` on the host.  Here every pass over the n x D descriptor
matrix runs on the GPU through the C-ABI (include/pvsim.h, "vocabulary training"); this module keeps what is K x D
sized -- the parameter update, the convergence tests, the random draws -- and follows the scikit-learn procedures it
replaces step by step (file:line given at each function; scikit-learn 1.7.2):

* same update formulas, stopping rules and defaults, so that with the same starting point the fitted tables agree with
  scikit-learn's to rounding (tests/test_gpu_parity.py compares against fits recorded in tests/golden/learn_*.npz);
* sums are formed in fp64 in a fixed order on the device (scikit-learn accumulates k-means sums in fp32 per thread
  chunk), and the data is not mean-centred first (KMeans.fit centres X only to lose fewer fp32 digits);
* random initialisation uses numpy's RandomState the way scikit-learn does but not draw-for-draw, so a seeded
  k-means++ start differs from scikit-learn's seeded start: statistically the same procedure, not the same sample.

There is no CPU path: without the HIP library these functions raise.
"""
from __future__ import annotations

import math
import warnings

import numpy as np

from . import _ffi
from .engine import DESC_F32, Context, default_context
from .models import GMMModel, KMeansModel, PCAModel

__all__ = ["DeviceRows", "fit_pca", "fit_kmeans", "fit_gmm", "kmeans_plusplus"]

_BIG_F32_BYTE = 0x7F  # 0x7f7f7f7f = 3.39e38: "no centre yet" for the running minimum distance


class DeviceRows:
    """(n, D) plain fp32 descriptor rows resident on the device."""

    def __init__(self, ctx: Context, n: int, D: int, buf):
        self.ctx, self.n, self.D, self.buf = ctx, int(n), int(D), buf

    @property
    def ptr(self) -> int:
        return self.buf.ptr

    @classmethod
    def from_host(cls, ctx: Context, x: np.ndarray, kind: int = DESC_F32) -> "DeviceRows":
        """Uploads (n, D) descriptors; `kind` says how the rows are stored (engine.DESC_*): the RootSIFT kinds are
        transformed on the device exactly as the encoders do."""
        x = np.ascontiguousarray(x)
        if x.ndim != 2 or x.shape[0] == 0:
            raise ValueError("descriptors must be a non-empty (n, D) array")
        want = np.uint8 if kind == _ffi.DESC_U8_ROOTSIFT else np.float32
        x = np.ascontiguousarray(x, dtype=want)
        n, D = x.shape
        out = ctx.buffer(n * D * 4)
        if kind == DESC_F32:
            out.upload(x)
        else:
            stage = ctx.buffer(x.nbytes).upload(x)
            ctx.materialise_dev(stage.ptr, kind, D, n, out.ptr)
            ctx.sync()
            stage.free()
        return cls(ctx, n, D, out)

    @classmethod
    def from_device(cls, ctx: Context, dptr: int, n: int, D: int) -> "DeviceRows":
        """Rows that already live on the device as contiguous fp32 (n, D), e.g. a torch tensor's data_ptr(); not owned."""
        from .engine import DeviceBuffer
        return cls(ctx, n, D, DeviceBuffer.view(ctx, dptr, int(n) * int(D) * 4))

    def row(self, i: int) -> np.ndarray:
        return self.buf.download((self.D,), np.float32, offset=int(i) * self.D * 4)

    def rows(self, idx) -> np.ndarray:
        return np.stack([self.row(i) for i in idx]) if len(idx) else np.zeros((0, self.D), np.float32)

    def transformed(self, pca: PCAModel) -> "DeviceRows":
        """PCA.transform of the rows, on the device (pvs_pca_transform_dev)."""
        table = self.ctx.pca(pca.components_, pca.mean_)
        out = self.ctx.buffer(self.n * pca.n_components * 4)
        self.ctx.pca_transform_dev(table, self.ptr, DESC_F32, self.n, out.ptr)
        self.ctx.sync()
        table.close()
        return DeviceRows(self.ctx, self.n, pca.n_components, out)

    def free(self):
        self.buf.free()


def _rng(random_state):
    """sklearn.utils.check_random_state"""
    if random_state is None or random_state is np.random:
        return np.random.mtrand._rand
    if isinstance(random_state, (int, np.integer)):
        return np.random.RandomState(random_state)
    if isinstance(random_state, np.random.RandomState):
        return random_state
    raise ValueError(f"{random_state!r} cannot be used to seed a numpy.random.RandomState instance")


def _one_blas_thread():
    """Context for the host's small dense algebra (a D x D eigen-decomposition): ONE BLAS / LAPACK thread.  A library pool of
    dozens of threads keeps spinning after such a call; inside a CPU-quota cgroup (the GPU boxes give a 256-CPU host 16 CPUs'
    worth) that burns the quota and the kernel throttles the whole process for the rest of its 100-ms period -- measured as an
    ~80-ms stall somewhere in the NEXT fit's launch chain (tests/tools/gap_trace.py: the host stops issuing HIP calls, once a
    hipLaunchKernel itself took 71 ms).  threadpoolctl is optional: without it the call runs as the library is configured."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except Exception:
        import contextlib
        return contextlib.nullcontext()


# ------------------------------------------------------------------------------------------------ PCA
def fit_pca(rows: DeviceRows, n_components: int) -> PCAModel:
    """PCA.fit with the "covariance_eigh" solver (sklearn/decomposition/_pca.py:_fit_full, the solver "auto" picks for
    n >= 10 D and D <= 1000, which is every descriptor matrix): mean, covariance from the Gram matrix (device, fp64),
    eigh, eigenvalues descending, svd_flip(u_based_decision=False), first n_components rows."""
    n, D = rows.n, rows.D
    if not 0 < n_components <= min(n, D):
        raise ValueError(f"n_components={n_components} must be between 1 and min(n_samples, n_features)={min(n, D)}")
    if n < 2:
        raise ValueError("PCA needs at least two descriptors")
    s, g = rows.ctx.gram_dev(rows.ptr, D, n)
    mean = s / n
    cov = (g - n * np.outer(mean, mean)) / (n - 1)
    with _one_blas_thread():
        vals, vecs = np.linalg.eigh(cov)
    vals, vecs = vals[::-1].copy(), vecs[:, ::-1]
    vals[vals < 0.0] = 0.0
    vt = vecs.T.copy()
    piv = np.argmax(np.abs(vt), axis=1)                       # svd_flip, v-based
    vt *= np.sign(vt[np.arange(D), piv])[:, None]
    model = PCAModel(vt[:n_components], mean)
    model.explained_variance_ = vals[:n_components].copy()
    total = vals.sum()
    model.explained_variance_ratio_ = model.explained_variance_ / total if total > 0 else np.zeros(n_components)
    model.singular_values_ = np.sqrt(vals[:n_components] * (n - 1))
    model.noise_variance_ = float(vals[n_components:].mean()) if n_components < min(n, D) else 0.0
    model.n_samples_, model.n_components_ = n, n_components
    return model


# ------------------------------------------------------------------------------------------------ k-means
def _draw_candidates(rows: DeviceRows, mind, block_sums: np.ndarray, r: np.ndarray, d_cand) -> np.ndarray:
    """searchsorted(cumsum(mind), r) (sklearn/cluster/_kmeans.py:_kmeans_plusplus, `candidate_ids`) without bringing mind to
    the host: the 4096-entry block of every draw from the per-block sums here, the position inside the block and the copy
    of the drawn rows into d_cand on the device (pvs_seed_pick_dev) -- one round trip per seeding step."""
    cum = np.cumsum(block_sums)
    blocks = np.minimum(np.searchsorted(cum, r), len(cum) - 1).astype(np.int64)
    base = np.where(blocks > 0, cum[np.maximum(blocks - 1, 0)], 0.0)
    return rows.ctx.seed_pick_dev(rows.ptr, rows.D, rows.n, mind.ptr, blocks, base, r, d_cand.ptr)


def kmeans_plusplus(rows: DeviceRows, n_clusters: int, random_state=None, n_local_trials=None, stepwise=False):
    """Greedy k-means++ (sklearn/cluster/_kmeans.py:_kmeans_plusplus): first centre uniform, every further centre the
    best of 2 + log(K) candidates drawn with probability proportional to the squared distance to the nearest chosen
    centre.  Draws, distances and potentials are device passes.  With up to 8 local trials (K < 404) the whole run is ONE
    call without a host round trip per step (pvs_kmeanspp_run_dev: the random numbers of all steps go to the device up front,
    targets = u * potential are formed there); `stepwise=True` takes the entry points one step at a time -- the same random
    stream, the same arithmetic, the same indices.  -> (centers (K, D) f32, indices (K,))"""
    ctx, n, D = rows.ctx, rows.n, rows.D
    if n_clusters > n:
        raise ValueError(f"n_samples={n} should be >= n_clusters={n_clusters}.")
    rng = _rng(random_state)
    trials = n_local_trials or 2 + int(math.log(n_clusters))
    if trials <= 8 and not stepwise:
        first = rng.randint(n)
        u = rng.uniform(size=(max(n_clusters - 1, 0), trials))      # row c-1: the draws of step c, as successive calls would give
        indices = ctx.kmeanspp_run_dev(rows.ptr, D, n, n_clusters, trials, u, first)
        return rows.rows(indices), indices
    indices = np.full(n_clusters, -1, dtype=np.int64)
    mind = ctx.buffer(n * 4).fill_bytes(_BIG_F32_BYTE)
    dist = ctx.buffer(min(trials, 8) * n * 4)
    cand = ctx.buffer(max(trials, 1) * D * 4)
    try:
        indices[0] = rng.randint(n)
        pot = float(ctx.seed_distances_dev(rows.ptr, D, n, rows.row(indices[0])[None], None, dist.ptr)[0])
        sums = ctx.min_update_dev(mind.ptr, dist.ptr, n)
        for c in range(1, n_clusters):
            ids = _draw_candidates(rows, mind, sums, rng.uniform(size=trials) * pot, cand)
            best = None                                         # (potential, descriptor index, group start, slot)
            for g0 in range(0, trials, 8):                      # the device scores up to 8 candidates per pass
                m = min(8, trials - g0)
                pots = ctx.seed_distances_dev(rows.ptr, D, n, cand.ptr + g0 * D * 4, mind.ptr, dist.ptr, n_cand=m)
                j = int(np.argmin(pots))
                if best is None or pots[j] < best[0]:
                    best = (float(pots[j]), int(ids[g0 + j]), g0, j)
            best_pot, best_id, g_best, best_slot = best
            if g_best != ((trials - 1) // 8) * 8:               # the winner's distances were overwritten: recompute them
                ctx.seed_distances_dev(rows.ptr, D, n, cand.ptr + (g_best + best_slot) * D * 4, mind.ptr, dist.ptr, n_cand=1)
                best_slot = 0
            sums = ctx.min_update_dev(mind.ptr, dist.ptr + best_slot * n * 4, n)
            pot = best_pot
            indices[c] = best_id
        centers = rows.rows(indices)
    finally:
        mind.free()
        dist.free()
        cand.free()
    return centers, indices


def _relocate_empty(rows, counts, sum_x, labels_buf, sqdist_buf):
    """sklearn/cluster/_k_means_common.pyx:_relocate_empty_clusters_dense: every empty cluster takes the descriptor
    that is farthest from its own centre (and that descriptor leaves its cluster's sum)."""
    empty = np.where(counts == 0)[0]
    if len(empty) == 0:
        return
    d = sqdist_buf.download((rows.n,), np.float32)
    far = np.argpartition(d, -len(empty))[:-len(empty) - 1:-1]
    for new_id, idx in zip(empty, far):
        x = rows.row(idx).astype(np.float64)
        old_id = int(labels_buf.download((1,), np.int32, offset=int(idx) * 4)[0])
        sum_x[old_id] -= x
        sum_x[new_id] = x
        counts[new_id] = 1
        counts[old_id] -= 1


def _lloyd(rows, centers, max_iter, tol_abs, verbose):
    """sklearn/cluster/_kmeans.py:_kmeans_single_lloyd"""
    ctx, n = rows.ctx, rows.n
    K, D = centers.shape
    lab = [ctx.buffer(n * 4), ctx.buffer(n * 4)]
    sqd = ctx.buffer(n * 4)
    try:
        cur, have_prev, strict, n_iter = 0, False, False, 0
        for i in range(max_iter):
            n_iter = i + 1
            cb = ctx.codebook(centers)
            resid, counts, inertia, changed = ctx.kmeans_step_dev(cb, rows.ptr, n, lab[cur].ptr,
                                                                  lab[1 - cur].ptr if have_prev else None, sqd.ptr)
            cb.close()
            if verbose:
                print(f"Iteration {i}, inertia {inertia}.")
            counts = counts.copy()
            sum_x = resid + counts[:, None] * centers.astype(np.float64)
            _relocate_empty(rows, counts, sum_x, lab[cur], sqd)
            new = centers.copy()
            nz = counts > 0
            new[nz] = (sum_x[nz] / counts[nz, None]).astype(np.float32)
            shift_tot = float(((new.astype(np.float64) - centers.astype(np.float64)) ** 2).sum())
            centers = new
            if have_prev and changed == 0:
                strict = True
                if verbose:
                    print(f"Converged at iteration {i}: strict convergence.")
                break
            if shift_tot <= tol_abs:
                if verbose:
                    print(f"Converged at iteration {i}: center shift {shift_tot} within tolerance {tol_abs}.")
                break
            cur, have_prev = 1 - cur, True
        # labels and inertia that belong to the final centres (the reference reruns the E-step unless strictly converged;
        # after strict convergence the centres moved by rounding at most, and the rerun makes the result self-consistent)
        cb = ctx.codebook(centers)
        _, counts, inertia, _ = ctx.kmeans_step_dev(cb, rows.ptr, n, lab[cur].ptr, None, None)
        cb.close()
        labels = lab[cur].download((n,), np.int32)
        return labels, inertia, centers, n_iter, counts
    finally:
        for b in lab:
            b.free()
        sqd.free()


def fit_kmeans(rows: DeviceRows, n_clusters: int, *, init="k-means++", n_init="auto", max_iter: int = 300,
               tol: float = 1e-4, random_state=None, verbose: int = 0, algorithm: str = "lloyd",
               copy_x: bool = True) -> KMeansModel:
    """KMeans(...).fit on the device (sklearn/cluster/_kmeans.py:KMeans.fit; defaults of scikit-learn 1.7)."""
    n, D = rows.n, rows.D
    if algorithm not in ("lloyd", "auto", "full", "elkan"):
        raise ValueError(f"unknown algorithm {algorithm!r}")
    if n < n_clusters:
        raise ValueError(f"n_samples={n} should be >= n_clusters={n_clusters}.")
    if max_iter < 1 or n_clusters < 1:
        raise ValueError("max_iter and n_clusters must be positive")
    rng = _rng(random_state)
    init_is_array = not isinstance(init, str) and not callable(init)
    if init_is_array:
        init = np.ascontiguousarray(init, dtype=np.float32)
        if init.shape != (n_clusters, D):
            raise ValueError(f"The shape of the initial centers {init.shape} does not match (n_clusters, n_features) = "
                             f"{(n_clusters, D)}.")
        runs = 1
    elif init in ("k-means++", "random"):
        runs = (1 if init == "k-means++" else 10) if n_init == "auto" else int(n_init)
    else:
        raise ValueError(f"init should be 'k-means++', 'random' or an array, got {init!r}")
    # tolerance relative to the data scale (sklearn _tolerance: mean of the per-feature variances * tol)
    s, g = rows.ctx.gram_dev(rows.ptr, D, n)
    var = np.maximum(np.diag(g) / n - (s / n) ** 2, 0.0)
    tol_abs = float(var.mean() * tol)
    best = None
    for _ in range(max(runs, 1)):
        if init_is_array:
            c0 = init.copy()
        elif init == "k-means++":
            c0, _idx = kmeans_plusplus(rows, n_clusters, rng)
        else:
            c0 = rows.rows(rng.permutation(n)[:n_clusters])
        labels, inertia, centers, n_iter, counts = _lloyd(rows, c0, max_iter, tol_abs, verbose)
        if best is None or inertia < best[1]:
            best = (labels, inertia, centers, n_iter, counts)
    labels, inertia, centers, n_iter, counts = best
    distinct = int((counts > 0).sum())
    if distinct < n_clusters:
        warnings.warn(f"Number of distinct clusters ({distinct}) found smaller than n_clusters ({n_clusters}). "
                      "Possibly due to duplicate points in X.", UserWarning, stacklevel=2)
    model = KMeansModel(centers)
    model.labels_, model.inertia_, model.n_iter_ = labels, float(inertia), int(n_iter)
    return model


# ------------------------------------------------------------------------------------------------ GMM
def _gmm_params_from_moments(s0, s1, s2, reg_covar):
    """sklearn/mixture/_gaussian_mixture.py:_estimate_gaussian_parameters + _estimate_gaussian_covariances_diag"""
    nk = s0 + 10 * np.finfo(np.float64).eps
    means = s1 / nk[:, None]
    cov = s2 / nk[:, None] - means ** 2 + reg_covar
    return nk, means, cov


def _check_cov(cov):
    if np.any(cov <= 0.0):                                   # _compute_precision_cholesky
        raise ValueError("Fitting the mixture model failed because some components have ill-defined empirical "
                         "covariance (for instance caused by singleton or collapsed samples). Try to decrease the "
                         "number of components, increase reg_covar, or scale the input data.")


def fit_gmm(rows: DeviceRows, n_components: int, *, tol: float = 1e-3, reg_covar: float = 1e-6, max_iter: int = 100,
            n_init: int = 1, init_params: str = "kmeans", weights_init=None, means_init=None, precisions_init=None,
            random_state=None, verbose: int = 0, covariance_type: str = "diag") -> GMMModel:
    """GaussianMixture(covariance_type="diag", ...).fit on the device (sklearn/mixture/_base.py:BaseMixture.fit_predict,
    _gaussian_mixture.py:_m_step / _initialize).  All arithmetic fp64."""
    if covariance_type != "diag":
        raise ValueError("only diagonal covariances are supported (the reference fixes covariance_type='diag')")
    ctx, n, D, K = rows.ctx, rows.n, rows.D, int(n_components)
    if n < 2 or n < K:
        raise ValueError(f"Expected n_samples >= n_components but got n_components = {K}, n_samples = {n}")
    if K > 1024:
        raise NotImplementedError("device GMM training supports at most 1024 components")
    rng = _rng(random_state)
    best = None
    for _ in range(max(int(n_init), 1)):
        # ---- initial parameters (BaseMixture._initialize_parameters -> GaussianMixture._initialize)
        if weights_init is not None and means_init is not None and precisions_init is not None:
            nk = means0 = cov0 = None                           # nothing of the data-driven start would be used
        elif init_params == "kmeans":
            km = fit_kmeans(rows, K, n_init=1, random_state=rng)
            lab = ctx.buffer(n * 4).upload(km.labels_)
            s1 = ctx.label_sums_dev(rows.ptr, D, n, lab.ptr, K, square=False)
            s2 = ctx.label_sums_dev(rows.ptr, D, n, lab.ptr, K, square=True)
            lab.free()
            s0 = np.bincount(km.labels_, minlength=K).astype(np.float64)
            nk, means0, cov0 = _gmm_params_from_moments(s0, s1, s2, reg_covar)
        elif init_params in ("k-means++", "random_from_data"):
            if init_params == "k-means++":
                _, idx = kmeans_plusplus(rows, K, rng)
            else:
                idx = rng.choice(n, size=K, replace=False)
            pts = rows.rows(idx).astype(np.float64)
            p2 = (rows.rows(idx) ** 2).astype(np.float64)       # X * X in X's dtype
            nk, means0, cov0 = _gmm_params_from_moments(np.ones(K), pts, p2, reg_covar)
        else:
            raise NotImplementedError(f"init_params={init_params!r} is not supported on the device")
        w = np.asarray(weights_init, np.float64) if weights_init is not None else nk / n
        mu = np.asarray(means_init, np.float64) if means_init is not None else means0
        cov = 1.0 / np.asarray(precisions_init, np.float64) if precisions_init is not None else cov0
        if mu.shape != (K, D) or cov.shape != (K, D) or w.shape != (K,):
            raise ValueError("initial GMM tables must be weights (K,), means (K, D), precisions (K, D)")
        _check_cov(cov)

        # ---- EM (BaseMixture.fit_predict main loop)
        lower, converged, n_iter = -np.inf, False, 0
        for n_iter in range(1, max_iter + 1):
            prev = lower
            g = ctx.gmm(w, mu, cov)
            s0, s1, s2, ll = ctx.gmm_em_step_dev(g, rows.ptr, n)
            g.close()
            nk, mu, cov = _gmm_params_from_moments(s0, s1, s2, reg_covar)
            w = nk / n
            w = w / w.sum()
            _check_cov(cov)
            lower = ll / n
            if verbose:
                print(f"  Iteration {n_iter}\t ll {lower:.5f}")
            if abs(lower - prev) < tol:
                converged = True
                break
        if best is None or lower > best[0]:
            best = (lower, w, mu, cov, converged, n_iter)
    lower, w, mu, cov, converged, n_iter = best
    if not converged and max_iter > 0:
        warnings.warn("Best performing initialization did not converge. Try different init parameters, or increase "
                      "max_iter, tol, or check for degenerate data.", UserWarning, stacklevel=2)
    model = GMMModel(w, mu, cov)
    model.precisions_cholesky_ = 1.0 / np.sqrt(cov)
    model.precisions_ = model.precisions_cholesky_ ** 2
    model.converged_, model.n_iter_, model.lower_bound_ = bool(converged), int(n_iter), float(lower)
    return model
