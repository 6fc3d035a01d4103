import os, sys
import numpy as np, torch
torch.cuda.init()
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle")]
import pvsim, pvsim_oracle as orc
from pvsim import learn
ctx = pvsim.Context(0)
for (K, D, n, seed) in [(5, 128, 300, 1), (5, 128, 300, 2), (5, 64, 300, 3), (16, 128, 300, 4), (5, 128, 5000, 5), (2, 128, 300, 6), (5, 30, 300, 7)]:
    rng = np.random.default_rng(seed)
    mu = rng.normal(0, 3, (K, D))
    x = (mu[rng.integers(0, K, n)] + rng.standard_normal((n, D))).astype(np.float32)
    c0 = x[rng.choice(n, K, replace=False)].copy()
    rows = learn.DeviceRows.from_host(ctx, x)
    lab = ctx.buffer(n * 4)
    cb = ctx.codebook(c0)
    resid, counts, inertia, changed = ctx.kmeans_step_dev(cb, rows.ptr, n, lab.ptr, None, None)
    labels = lab.download((n,), np.int32)
    ref_l = orc.kmeans_predict(x, c0)
    r = np.zeros((K, D)); np.add.at(r, ref_l, x.astype(np.float64) - c0[ref_l].astype(np.float64))
    print(K, D, n, "labels equal", np.array_equal(labels, ref_l), "counts equal", np.array_equal(counts, np.bincount(ref_l, minlength=K)),
          "resid max diff", float(np.abs(resid - r).max()), "inertia rel", abs(inertia - ((x - c0[ref_l]) ** 2).sum()) / inertia)
