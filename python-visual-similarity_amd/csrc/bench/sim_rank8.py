"""Per-stage time of ONE rank's retrieval work at P = 8 (8189-image corpus), all on one GPU: where do the 3.3 ms go?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pvsim
from pvsim import distributed as pd
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev); torch.cuda.set_stream(side)
ctx = pvsim.Context(0, stream=side.cuda_stream)
N, L, k, P = 8189, 32768, 5, 8
g = torch.Generator(device=dev); g.manual_seed(1)
_, _, B = pd.shard_range(N, P, 0)
enc_all = torch.randn((P * B, L), generator=g, device=dev) * (torch.rand((P * B, L), generator=g, device=dev) > 0.4)
inv_all = torch.empty((P * B,), device=dev)
ctx.row_inv_norms_dev(enc_all.data_ptr(), P * B, L, inv_all.data_ptr())
ops = pd.DeviceOps(ctx, same_stream=True)
nt = lambda shape, dtype, fill: torch.full(shape, fill, dtype=dtype, device=dev)
for rank in (0, 3):
    for it in range(3):
        torch.cuda.synchronize(); ctx.timers_enable(True); ctx.timers_reset(); t0 = time.perf_counter()
        st = pd.symmetric_local(enc_all, inv_all, N, rank, P, k, ops, nt)
        i_, v_ = pd.symmetric_finish(st, ops, nt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tm = ctx.timers(); ctx.timers_enable(False)
    print(f"rank {rank}: wall {dt*1e3:.2f} ms; " + ", ".join(f"{n} {v[0]:.2f} ms / {v[1]} launches" for n, v in tm.items() if v[1]), flush=True)
