import os, sys, warnings
import numpy as np, torch
torch.cuda.init()
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle")]
import pvsim, pvsim_oracle as orc
from pvsim import learn
warnings.simplefilter("ignore")
ctx = pvsim.Context(0)
K, D, n = 40, 128, 300
found = 0
for seed in range(400):
    rng = np.random.default_rng(seed)
    mu = rng.normal(0, 3, (K, D))
    x = (mu[rng.integers(0, K, n)] + rng.standard_normal((n, D))).astype(np.float32)
    c0 = x[rng.choice(n, K, replace=False)].copy()
    rows = learn.DeviceRows.from_host(ctx, x)
    for it in (1, 2, 3):
        m = learn.fit_kmeans(rows, K, init=c0, n_init=1, max_iter=it, tol=0.0)
        rc, rl, _, rn = orc.kmeans_lloyd(x, c0, max_iter=it, tol=0.0, center=False)
        mis = int((m.labels_ != rl).sum())
        if mis > 10:
            print("seed", seed, "iters", it, "mismatches", mis, "device n_iter", m.n_iter_, "restatement n_iter", rn,
                  "device counts", np.bincount(m.labels_, minlength=K), "restatement counts", np.bincount(rl, minlength=K),
                  "max centre diff", float(np.abs(m.cluster_centers_ - rc).max()), flush=True)
            found += 1
            break
    rows.free()
    if found >= 2:
        break
print("done, found", found)
