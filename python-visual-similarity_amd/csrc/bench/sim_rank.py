import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pvsim
from pvsim import distributed as pd
dev = torch.device("cuda", 0)
ctx = pvsim.Context(0)
N, L, k = 8189, 32768, 5
g = torch.Generator(device=dev); g.manual_seed(1)
for P in (1, 2, 4, 8):
    _, _, B = pd.shard_range(N, P, 0)
    enc_all = torch.randn((P * B, L), generator=g, device=dev) * (torch.rand((P * B, L), generator=g, device=dev) > 0.4)
    inv_all = torch.empty((P * B,), device=dev)
    torch.cuda.synchronize()
    ctx.row_inv_norms_dev(enc_all.data_ptr(), P * B, L, inv_all.data_ptr()); ctx.sync()
    ops = pd.DeviceOps(ctx)
    def nt(shape, dtype, fill):
        t = torch.full(shape, fill, dtype=dtype, device=dev); torch.cuda.synchronize(); return t
    times = []
    for rank in (0, P - 1):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = pd.symmetric_local(enc_all, inv_all, N, rank, P, k, ops, nt)
            i_, v_ = pd.symmetric_finish(st, ops, nt)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        times.append(dt * 1e3)
    # the simple scheme for comparison: one launch B x N
    idx = torch.empty((B, k), dtype=torch.int64, device=dev); val = torch.empty((B, k), device=dev)
    torch.cuda.synchronize()
    for it in range(2):
        t0 = time.perf_counter()
        ctx.cosine_topk_dev(enc_all.data_ptr(), min(B, N), enc_all.data_ptr(), N if P > 1 else min(B, N), L, inv_all.data_ptr(), inv_all.data_ptr(), k, 0, False, idx.data_ptr(), val.data_ptr())
        ctx.sync(); ds = time.perf_counter() - t0
    print(f"P={P} B={B}: symmetric scheme retrieve per rank: rank0 {times[0]:.2f} ms, last rank {times[1]:.2f} ms; one-launch B x N: {ds*1e3:.2f} ms", flush=True)
    del enc_all
