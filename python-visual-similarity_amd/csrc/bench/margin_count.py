"""How many database columns lie within a given margin of each query's k-th best cosine score (bench corpus)?
Sizing experiment for a prefilter + exact re-scoring retrieval; not part of the product."""
import os, sys
import numpy as np
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "python-visual-similarity_amd"))
import bench, pvsim
from pvsim.engine import DESC_F32
dev = torch.device("cuda", 0)
ctx = pvsim.Context(0)
tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"))
cb = ctx.codebook(tables["centroids"])
N = 8189
raw, offsets = bench.make_corpus(N, 1235, dev)
desc = bench.rootsift_torch(raw)
d_off = torch.from_numpy(offsets).to(dev)
enc = torch.empty((N, 32768), dtype=torch.float32, device=dev)
inv = torch.empty((N,), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx.vlad_encode_dev(cb, desc.data_ptr(), DESC_F32, d_off.data_ptr(), N, int(offsets[-1]), enc.data_ptr(), d_inv_norm=inv.data_ptr())
S = torch.empty((N, N), dtype=torch.float32, device=dev)
ctx.cosine_dev(enc.data_ptr(), N, enc.data_ptr(), N, 32768, inv.data_ptr(), inv.data_ptr(), S.data_ptr(), N)
ctx.sync()
e16 = enc.half().float()
for k in (5, 100):
    kth = torch.topk(S, k, dim=1).values[:, -1:]
    for m in (5e-4, 2.2e-3, 6e-3, 2e-2):
        c = (S >= kth - m).sum(1).float()
        print(f"k={k} margin {m:g}: candidates per row mean {c.mean():.1f} median {c.median():.0f} p99 {c.quantile(0.99):.0f} max {c.max():.0f}")
# observed fp16 error on a sample of pairs
idx = torch.randint(0, N, (2000, 2), device=dev)
a, b = idx[:, 0], idx[:, 1]
s32 = S[a, b]
s16 = (e16[a] * e16[b]).sum(1) * inv[a] * inv[b]
print("observed |fp16-input score - fp32 score| max", float((s16 - s32).abs().max()), "score std", float(S.std()), "mean", float(S.mean()))
