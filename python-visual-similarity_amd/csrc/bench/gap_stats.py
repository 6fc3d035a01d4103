"""Distribution of the gap between the best and second-best |c|^2 - 2 x.c over the bench corpus (sizing experiment)."""
import os, sys
import numpy as np, torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "python-visual-similarity_amd"))
import bench
dev = torch.device("cuda", 0)
tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"))
C = torch.from_numpy(tables["centroids"]).to(dev)
raw, offsets = bench.make_corpus(1024, 1235, dev)
x = bench.rootsift_torch(raw)[:1000000]
v = (C * C).sum(1)[None, :] - 2.0 * (x.double() @ C.double().T)
two = torch.topk(v, 2, dim=1, largest=False).values
gap = (two[:, 1] - two[:, 0]).float()
print("cmax", float(C.norm(dim=1).max()), "nx mean", float(x.norm(dim=1).mean()))
for m in (4e-3, 1e-3, 2.3e-4, 1e-4, 2e-5):
    print(f"margin {m:g}: fraction of descriptors with gap < margin: {float((gap < m).float().mean()):.4f}")
print("gap quantiles", [float(gap.quantile(q)) for q in (0.01, 0.1, 0.5, 0.9)])
