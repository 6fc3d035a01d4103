import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "python-visual-similarity_amd"))
import pvsim
from pvsim.engine import DESC_F32
ctx = pvsim.Context(0)
K, D, n, N = 256, 512, 196, 8189
rng = np.random.default_rng(1)
gm = ctx.gmm(rng.dirichlet(np.full(K, 5.0)), rng.normal(0, 2, (K, D)), np.exp(rng.uniform(-3, 3, (K, D))))
dev = torch.device("cuda", 0)
desc = torch.randn((N * n, D), device=dev)
off = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * n
enc = torch.empty((N, K + 2 * K * D), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for it in range(3):
    ctx.timers_enable(True); ctx.timers_reset()
    ctx.fisher_encode_dev(gm, desc.data_ptr(), DESC_F32, off.data_ptr(), N, N * n, enc.data_ptr(), 0)
    ctx.sync()
    t = ctx.timers()
print("PVS_MOM_DBG", os.environ.get("PVS_MOM_DBG"), {k: round(v[0], 3) for k, v in t.items() if v[1]}, flush=True)
