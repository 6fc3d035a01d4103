#!/usr/bin/env python3
"""Phase breakdown of the K1 prefilter (assign16_kernel, in-kernel s_memtime stamps of waves 0 and 4, diagnostic build):
python tests/tools/assign_profile.py [n_images] [u8|f32]"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "python-visual-similarity_amd"))
import torch
import pvsim
from pvsim import synth
from pvsim.engine import DESC_F32, DESC_U8_ROOTSIFT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
u8 = len(sys.argv) > 2 and sys.argv[2] == "u8"
n = 512
dev = torch.device("cuda", 0)
ctx = pvsim.Context(0)
t = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"), allow_pickle=False)
cb = ctx.codebook(t["centroids"])
from pvsim import _ffi
two = {}
g = torch.Generator(device=dev); g.manual_seed(1)
proto = torch.from_numpy(synth.sift_prototypes().astype(np.float32)).to(dev)
raw = torch.empty((N * n, 128), dtype=torch.uint8, device=dev)
for s0 in range(0, N * n, 1 << 20):
    e0 = min(N * n, s0 + (1 << 20))
    z = torch.randint(0, proto.shape[0], (e0 - s0,), generator=g, device=dev)
    x = proto[z] * torch.exp(0.35 * torch.randn((e0 - s0, 128), generator=g, device=dev)) + 4.8 * torch.rand((e0 - s0, 128), generator=g, device=dev) ** 3
    raw[s0:e0] = (x * (512.0 / x.norm(dim=1, keepdim=True).clamp_min(1e-9))).clamp_max(255.0).round().to(torch.uint8)
if u8:
    desc, kind = raw, DESC_U8_ROOTSIFT
else:
    xf = raw.float(); desc, kind = torch.sqrt(xf / (xf.sum(dim=1, keepdim=True) + 1e-7)).contiguous(), DESC_F32
off = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * n
enc = torch.empty((N, 32768), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
def run():
    ctx.vlad_encode_dev(cb, desc.data_ptr(), kind, off.data_ptr(), N, N * n, enc.data_ptr())
for _ in range(2): run()
ctx.sync()
ctx.timers_enable(True); ctx.timers_reset()
for _ in range(3): run()
ctx.sync()
tm = ctx.timers(); ctx.timers_enable(False)
ctx.fused_profile(True)
run(); ctx.sync()
st = ctx.fused_profile(False, raw=True)
blocks = max(int(st[10]), 1)
names = ("conversion", "barrier_before_mfma", "mfma_loop", "barrier_after_mfma", "tail")
out = {"images": N, "kind": "u8" if u8 else "f32", "assign_ms": round(tm["assign"][0] / max(tm["assign"][1], 1), 3),
       "aggregate_ms": round(tm["aggregate"][0] / max(tm["aggregate"][1], 1), 3), "blocks_of_256_rows": blocks,
       "cycles_per_block_wave0": {k: round(int(st[i]) / blocks, 1) for i, k in enumerate(names)},
       "cycles_per_block_wave4": {k: round(int(st[5 + i]) / blocks, 1) for i, k in enumerate(names)},
       "mfma_cycles_per_block_per_wave": 200 * 32}
print(json.dumps(out))
