import os, sys, time
import numpy as np, torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "python-visual-similarity_amd"))
import bench, pvsim
from pvsim.engine import DESC_F32
dev = torch.device("cuda", 0)
ctx = pvsim.Context(0)
tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"))
cb = ctx.codebook(tables["centroids"])
raw, offsets = bench.make_corpus(8189, 1235, dev)
desc = bench.rootsift_torch(raw)
n = desc.shape[0]
lab = torch.empty((n,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for flag in ("0", "1", "2", "3", "4", "7"):
    os.environ["PVS_A16_DBG"] = flag
    for it in range(3):
        ctx.sync(); t0 = time.perf_counter()
        ctx.kmeans_predict_dev(cb, desc.data_ptr(), DESC_F32, n, lab.data_ptr())
        ctx.sync(); dt = time.perf_counter() - t0
    print("dbg", flag, "total assign ms", round(dt * 1e3, 3))
