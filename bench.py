#!/usr/bin/env python3
"""bench.py -- images/sec encoded + top-k retrieved, VLAD K=256 RootSIFT (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over the whole synthetic corpus, inputs already resident in HBM:
  encode  : descriptors -> KMeans assignment -> VLAD residual aggregation -> power + intra L2 norm
  exchange: (N > 1) RCCL all-gather of the per-GPU encoding blocks
  retrieve: every image queries the whole corpus: N x N cosine GEMM + top-k (k = 5)
Workload at 1 GPU = BASELINE.json configs[1]: 8189 images (Oxford-102 sized), ragged descriptor counts
(LogNormal around 1257), D = 128, K = 256.  With N GPUs the same corpus is sharded by image
(strong scaling); each rank scores its own query block against the gathered corpus.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "python-visual-similarity_amd"))

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
K_CLUSTERS, DIM, TOPK = 256, 128, 5


def _gemm_is_f16(name):
    """gemm_mfma_kernel<BM, BN, WM, WN, STAGES, SYMM, OCC, MODE, STAMP, F16, ...>: the 10th template argument"""
    try:
        return name.split("<", 1)[1].split(",")[9].strip() == "true"
    except IndexError:
        return False


def pmc_mfma_busy(kernel_prefix, keep=lambda name: True):
    """MFMA pipe utilisation of a kernel from the newest committed PMC summary (launch-weighted over its variants)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm.json")))
    try:
        data = json.load(open(files[-1]))
        v = [d["mfma_pipe_busy_frac"] for name, d in data["kernels"].items()
             if name.startswith(kernel_prefix) and keep(name) and "mfma_pipe_busy_frac" in d]
        return max(v) if v else None      # the main launch (the split-K partial / reduce launches are tiny)
    except Exception:
        return None


def pmc_traffic(kernel_prefix, keep=lambda name: True):
    """HBM-side bytes per launch of a kernel from the newest committed PMC summary (profiles/rNN_pmc_hbm.json,
    produced by profiles/summarize.py from separate rocprofv3 --pmc passes of this same command)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm.json")))
    if not files:
        return None, None
    try:
        data = json.load(open(files[-1]))
        tot = [d["hbm_bytes_per_launch_corrected"] for name, d in data["kernels"].items()
               if name.startswith(kernel_prefix) and keep(name) and "hbm_bytes_per_launch_corrected" in d]
        if tot:   # main + split-K partial + reduce launches together are one GEMM step
            return float(sum(tot)), os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=8189)
    ap.add_argument("--desc", choices=["f32", "u8"], default="f32",
                    help="descriptor rows in HBM: fp32 RootSIFT (default) or raw uint8 SIFT with fused RootSIFT")
    ap.add_argument("--workload", choices=["config2", "fisher", "vlad512", "fp16sim", "learn", "corpus1m"], default="config2",
                    help="config2 = the headline line (default).  Side workloads (single GPU, same JSON shape, not the "
                         "headline): fisher = BASELINE configs[2] (Fisher D=512 K=256 n=196), vlad512 = the per-GPU share of "
                         "configs[3] (n=512 descriptors per image, encode only), fp16sim = configs[4] scaled to one GPU "
                         "(N x N cosine on fp16 encodings + top-10)")
    ap.add_argument("--retrieval", choices=["exact", "filtered"], default="exact",
                    help="exact = f32 MFMA GEMM over all pairs (the headline). filtered = the same top-k lists, bit for bit, "
                         "through the fp16 prefilter + exact re-scoring (pvs_cosine_topk_filtered_dev); single GPU only. The "
                         "default run also times the filtered variant and reports it under 'filtered_retrieval'.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of cpu_baseline_subprocess()
    ap.add_argument("--pcie", action="store_true", help="also time one step with host-resident inputs (H2D included)")
    return ap.parse_args()


def make_corpus(n_images, seed, device):
    """Synthetic SIFT-like descriptors generated ON DEVICE (torch is plumbing here): a prototype histogram
    times log-normal noise plus a sparse floor, scaled to L2 = 512, clipped to 255, rounded -> uint8."""
    import torch
    from pvsim import synth
    counts = synth.ragged_counts(n_images, seed)
    offsets = np.zeros(n_images + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    total = int(offsets[-1])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    proto = torch.from_numpy(synth.sift_prototypes().astype(np.float32)).to(device)
    raw = torch.empty((total, DIM), dtype=torch.uint8, device=device)
    step = 1 << 20
    for s in range(0, total, step):
        e = min(total, s + step)
        z = torch.randint(0, proto.shape[0], (e - s,), generator=g, device=device)
        x = proto[z] * torch.exp(0.35 * torch.randn((e - s, DIM), generator=g, device=device))
        x = x + 1.2 * torch.rand((e - s, DIM), generator=g, device=device) ** 3 * 4.0
        x = x * (512.0 / x.norm(dim=1, keepdim=True).clamp_min(1e-9))
        raw[s:e] = x.clamp_max(255.0).round().to(torch.uint8)
    return raw, offsets


def rootsift_torch(raw_u8):
    """fp32 RootSIFT rows, same arithmetic as the reference extractor tail (features/_features.py:112-114)."""
    import torch
    out = torch.empty(raw_u8.shape, dtype=torch.float32, device=raw_u8.device)
    step = 1 << 20
    for s in range(0, raw_u8.shape[0], step):
        x = raw_u8[s:s + step].float()
        out[s:s + step] = torch.sqrt(x / (x.sum(dim=1, keepdim=True) + 1e-7))
    return out


def side_workload(args):
    """Single-GPU measurements of the other BASELINE configs (same timing discipline, separate JSON line)."""
    import torch
    import pvsim
    from pvsim.engine import DESC_F32, DESC_U8_ROOTSIFT
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = pvsim.Context(0)
    tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"), allow_pickle=False)
    g = torch.Generator(device=dev)
    g.manual_seed(1236)
    N = args.images
    out = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "data": "synthetic",
           "vs_baseline": None, "device": ctx.device_name()}

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        ctx.sync(); torch.cuda.synchronize()
        ctx.timers_enable(True); ctx.timers_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        ctx.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        tm = ctx.timers(); ctx.timers_enable(False)
        return dt, {k: round(v[0] / args.steps, 4) for k, v in tm.items() if v[1]}

    if args.workload == "fisher":
        K, D, n = 256, 512, 196
        rng = np.random.default_rng(1236)
        means = rng.normal(0.0, 2.0, size=(K, D))
        cov = np.exp(rng.uniform(np.log(1e-3), np.log(25.0), size=(K, D)))
        w = rng.dirichlet(np.full(K, 5.0))
        gm = ctx.gmm(w, means, cov)
        z = torch.randint(0, K, (N * n,), generator=g, device=dev)
        desc = (torch.from_numpy(means.astype(np.float32)).to(dev)[z]
                + torch.from_numpy(np.sqrt(cov).astype(np.float32)).to(dev)[z] * torch.randn((N * n, D), generator=g, device=dev))
        off = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * n
        L = K + 2 * K * D
        enc = torch.empty((N, L), dtype=torch.float32, device=dev)
        inv = torch.empty((N,), dtype=torch.float32, device=dev)
        idx = torch.empty((N, TOPK), dtype=torch.int64, device=dev)
        val = torch.empty((N, TOPK), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        chunk = 32768

        def step():
            for s0 in range(0, N, chunk):
                e0 = min(N, s0 + chunk)
                ctx.fisher_encode_dev(gm, desc[s0 * n:].data_ptr(), DESC_F32, (off[s0:e0 + 1] - off[s0]).contiguous().data_ptr(),
                                      e0 - s0, (e0 - s0) * n, enc[s0:].data_ptr(), 0)
            ctx.row_inv_norms_dev(enc.data_ptr(), N, L, inv.data_ptr())
            if args.retrieval == "filtered":
                fstat[0] = ctx.cosine_topk_filtered_dev(enc.data_ptr(), N, enc.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), TOPK,
                                                        idx.data_ptr(), val.data_ptr())
            else:
                ctx.cosine_topk_dev(enc.data_ptr(), N, enc.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), TOPK, 0, False,
                                    idx.data_ptr(), val.data_ptr())
        fstat = [None]
        dt, st = timed(step)
        assert np.array_equal(idx[:, 0].cpu().numpy(), np.arange(N))
        out["retrieval"] = args.retrieval
        if fstat[0]:
            out["filtered_stats"] = fstat[0]
        enc_ms = st.get("fisher_posterior", 0) + st.get("fisher_moments", 0)
        out.update({"metric": "images/sec encoded + top-k retrieved, Fisher K256 D512 n196 (BASELINE configs[2])",
                    "value": round(N / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "f64",
                    "scaling": "strong", "stages_ms_per_step": st,
                    "config": {"workload": f"{N} images x {n} x {D}-D descriptors, diag GMM K={K}: Fisher encode (fp64 "
                                           f"arithmetic, fp32 stored) + {N}x{N} cosine + top-{TOPK}", "K": K, "D": D},
                    "encode_GFLOPs_fp64": round(8.0 * n * K * D * N / (enc_ms * 1e-3) / 1e9, 1) if enc_ms else None})
    elif args.workload == "vlad512":
        n = 512
        cb = ctx.codebook(tables["centroids"])
        from pvsim import synth
        proto = torch.from_numpy(synth.sift_prototypes().astype(np.float32)).to(dev)
        raw = torch.empty((N * n, DIM), dtype=torch.uint8, device=dev)
        for s0 in range(0, N * n, 1 << 20):
            e0 = min(N * n, s0 + (1 << 20))
            z = torch.randint(0, proto.shape[0], (e0 - s0,), generator=g, device=dev)
            x = proto[z] * torch.exp(0.35 * torch.randn((e0 - s0, DIM), generator=g, device=dev)) + \
                4.8 * torch.rand((e0 - s0, DIM), generator=g, device=dev) ** 3
            raw[s0:e0] = (x * (512.0 / x.norm(dim=1, keepdim=True).clamp_min(1e-9))).clamp_max(255.0).round().to(torch.uint8)
        desc = raw if args.desc == "u8" else rootsift_torch(raw)
        kind = DESC_U8_ROOTSIFT if args.desc == "u8" else DESC_F32
        off = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * n
        L = K_CLUSTERS * DIM
        enc = torch.empty((N, L), dtype=torch.float32, device=dev)
        inv = torch.empty((N,), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        dt, st = timed(lambda: ctx.vlad_encode_dev(cb, desc.data_ptr(), kind, off.data_ptr(), N, N * n, enc.data_ptr(),
                                                   d_inv_norm=inv.data_ptr()))
        byt = N * (n * DIM * (1 if args.desc == "u8" else 4) + L * 4)
        out.update({"metric": "images/sec VLAD-encoded, K256 D128 n512 (per-GPU share of BASELINE configs[3])",
                    "value": round(N / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "f32",
                    "scaling": "weak", "stages_ms_per_step": st,
                    "config": {"workload": f"{N} images x 512 SIFT-like descriptors ({args.desc}), VLAD K=256 encode only"},
                    # the whole encode (assign + aggregate) against HBM: descriptors read twice (K1, K2) is what the kernels do,
                    # the algorithmic bytes count them once (393,216 B / image from f32 rows)
                    "roofline": {"kernel": "assign16_kernel + assign_kernel (near ties) + vlad_aggregate_kernel", "bound": "hbm",
                                 "achieved": round(byt / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(byt / dt / 1e9 / 8000.0, 4), "traffic": None,
                                 "assign_executed_f16_TFLOPs": round(3 * 2.0 * N * n * K_CLUSTERS * DIM / (st["assign"] * 1e-3) / 1e12, 1)},
                    "encode_algorithmic_GBps": round(byt / dt / 1e9, 1)})
    elif args.workload == "corpus1m":
        # BASELINE configs[3]/[4] at their real size on ONE GPU: N images x 512 raw SIFT-like uint8 descriptors are generated
        # on the device chunk by chunk, VLAD-encoded (fused RootSIFT) and kept as fp16 rows with fp32 1/||.|| (65.5 GB at
        # N = 1e6: the corpus the 8-GPU configuration shards); then 8192 queries rank against the whole corpus (fp16 MFMA
        # GEMM, fp32 accumulate, fused top-10).  One pass, no warm-up repetitions of the corpus build.
        from pvsim import synth
        n, k, L, CH = 512, 10, K_CLUSTERS * DIM, 16384
        cb = ctx.codebook(tables["centroids"])
        proto = torch.from_numpy(synth.sift_prototypes().astype(np.float32)).to(dev)
        exact_mode = args.retrieval == "filtered"      # keep the fp32 corpus (4 bytes / element) and rank EXACTLY through the filter
        db16 = None if exact_mode else torch.empty((N, L), dtype=torch.float16, device=dev)
        db32 = torch.empty((N, L), dtype=torch.float32, device=dev) if exact_mode else None
        inv = torch.empty((N,), dtype=torch.float32, device=dev)
        enc = torch.empty((CH, L), dtype=torch.float32, device=dev)
        off = (torch.arange(CH + 1, device=dev, dtype=torch.int64) * n).contiguous()
        raw = torch.empty((CH * n, DIM), dtype=torch.uint8, device=dev)
        ctx.timers_enable(True); ctx.timers_reset()
        t_gen = t_enc = 0.0
        for c0 in range(0, N, CH):
            cn = min(CH, N - c0)
            t0 = time.perf_counter()
            for s0 in range(0, cn * n, 1 << 21):
                e0 = min(cn * n, s0 + (1 << 21))
                z = torch.randint(0, proto.shape[0], (e0 - s0,), generator=g, device=dev)
                x = proto[z] * torch.exp(0.35 * torch.randn((e0 - s0, DIM), generator=g, device=dev)) + \
                    4.8 * torch.rand((e0 - s0, DIM), generator=g, device=dev) ** 3
                raw[s0:e0] = (x * (512.0 / x.norm(dim=1, keepdim=True).clamp_min(1e-9))).clamp_max(255.0).round().to(torch.uint8)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.vlad_encode_dev(cb, raw.data_ptr(), DESC_U8_ROOTSIFT, off.data_ptr(), cn, cn * n,
                                (db32[c0:] if exact_mode else enc).data_ptr(), d_inv_norm=inv[c0:].data_ptr())
            if not exact_mode:
                ctx.f32_to_f16_dev(enc.data_ptr(), cn * L, db16[c0:].data_ptr())
            ctx.sync()
            t2 = time.perf_counter()
            t_gen += t1 - t0
            t_enc += t2 - t1
            if (c0 // CH) % 8 == 0:
                print(f"[corpus1m] {c0 + cn} / {N} images encoded", file=sys.stderr, flush=True)
        tm_enc = ctx.timers()
        nq = min(8192, N)
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        val = torch.empty((nq, k), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.timers_reset()
        fst = None
        if exact_mode:     # first call: the 2-byte workspace copy of the corpus is allocated (hipMalloc of N*L*2 bytes); timed apart
            t0 = time.perf_counter()
            ctx.cosine_topk_filtered_dev(db32.data_ptr(), nq, db32.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), k,
                                         idx.data_ptr(), val.data_ptr())
            ctx.sync()
            out["first_call_with_workspace_allocation_s"] = round(time.perf_counter() - t0, 3)
            ctx.timers_reset()
        t0 = time.perf_counter()
        if exact_mode:
            fst = ctx.cosine_topk_filtered_dev(db32.data_ptr(), nq, db32.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), k,
                                               idx.data_ptr(), val.data_ptr())
        else:
            ctx.cosine_topk_f16_dev(db16.data_ptr(), nq, db16.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), k, 0, False,
                                    idx.data_ptr(), val.data_ptr())
        ctx.sync()
        t_ret = time.perf_counter() - t0
        tm_ret = ctx.timers(); ctx.timers_enable(False)
        if exact_mode:
            # the all-pairs f32 GEMM over the same corpus: must give the same lists, bit for bit
            xi, xv = torch.empty_like(idx), torch.empty_like(val)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.cosine_topk_dev(db32.data_ptr(), nq, db32.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), k, 0, False,
                                xi.data_ptr(), xv.data_ptr())
            ctx.sync()
            t_exact = time.perf_counter() - t0
            same_lists = bool(torch.equal(xi, idx)) and bool(torch.equal(xv.view(torch.int32), val.view(torch.int32)))
            assert same_lists, "filtered retrieval differs from the all-pairs f32 GEMM"
            out["exact_check"] = {"all_pairs_f32_seconds": round(t_exact, 3), "lists_bit_identical": same_lists, "filter_stats": fst}
        assert np.array_equal(idx[:, 0].cpu().numpy(), np.arange(nq)), "self-retrieval failed"
        flop = 2.0 * nq * N * L
        out.update({"metric": "images/sec VLAD-encoded into a resident fp16 corpus, then query rows/sec against it (BASELINE configs[3]/[4] on one GPU)",
                    "value": round(N / t_enc, 1), "unit": "images/s", "ms_per_step": round(t_enc * 1e3, 1), "steps": 1, "warmup": 0,
                    "dtype": "f32 encode, f16 retrieval operands", "scaling": "weak",
                    "config": {"workload": f"{N} images x {n} uint8 descriptors -> {'fp32' if exact_mode else 'fp16'} corpus "
                                           f"({N * L * (4 if exact_mode else 2) / 1e9:.1f} GB resident); "
                                           f"{nq} queries x {N} rows, top-{k}"},
                    "corpus_build": {"encode_s": round(t_enc, 3), "generate_s": round(t_gen, 1),
                                     "stages_ms": {kk: round(v[0], 1) for kk, v in tm_enc.items() if v[1]}},
                    "retrieval": {"seconds": round(t_ret, 3), "queries_per_s": round(nq / t_ret, 1),
                                  "algorithmic_TFLOPs": round(flop / t_ret / 1e12, 1),
                                  "stages_ms": {kk: round(v[0], 1) for kk, v in tm_ret.items() if v[1]}},
                    "hbm_resident_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)})
    elif args.workload == "learn":
        # vocabulary training: one "step" = k-means++ seeding + 10 Lloyd iterations (K=256)
        # and 5 EM iterations of a K=256 diagonal GMM over args.images x 64 RootSIFT descriptors
        from pvsim import learn
        n_img = min(N, 16384)
        raw, offsets = make_corpus(n_img, 1238, dev)
        total = min(int(offsets[-1]), n_img * 64)
        x = rootsift_torch(raw[:total]).contiguous()
        rows = learn.DeviceRows.from_device(ctx, x.data_ptr(), total, DIM)
        torch.cuda.synchronize()
        res = {}

        def one():
            t0 = time.perf_counter()
            c0, _ = learn.kmeans_plusplus(rows, K_CLUSTERS, random_state=0)
            t1 = time.perf_counter()
            km = learn.fit_kmeans(rows, K_CLUSTERS, init=c0, n_init=1, max_iter=10, tol=0.0)
            t2 = time.perf_counter()
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                gm = learn.fit_gmm(rows, K_CLUSTERS, weights_init=np.full(K_CLUSTERS, 1.0 / K_CLUSTERS), means_init=km.cluster_centers_,
                                   precisions_init=np.full((K_CLUSTERS, DIM), 1.0 / 2e-3), max_iter=5, tol=0.0)
            t3 = time.perf_counter()
            pca = learn.fit_pca(rows, 64)
            t4 = time.perf_counter()
            res.update(seeding_s=t1 - t0, lloyd10_s=t2 - t1, em5_s=t3 - t2, pca_s=t4 - t3, inertia=km.inertia_, lower_bound=gm.lower_bound_)

        dt, st = timed(one)
        out.update({"metric": "descriptors/sec through k-means++ seeding + 10 Lloyd iterations + 5 EM iterations + PCA fit (K=256, D=128)",
                    "value": round(total / dt, 1), "unit": "descriptors/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "f32 (k-means), f64 (EM, PCA)",
                    "scaling": "strong", "stages_ms_per_step": st, "phases_s": {k: round(v, 4) for k, v in res.items()},
                    "config": {"workload": f"{total} RootSIFT descriptors x {DIM}, K = {K_CLUSTERS}"}})
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import pvsim_oracle as orc
            sub = x[:100000].cpu().numpy()
            c0 = sub[:K_CLUSTERS].copy()
            t0 = time.perf_counter()
            orc.kmeans_lloyd(sub, c0, max_iter=2, tol=0.0)
            t_l = (time.perf_counter() - t0) / 2 / len(sub)
            t0 = time.perf_counter()
            orc.gmm_em(sub, np.full(K_CLUSTERS, 1.0 / K_CLUSTERS), c0, np.full((K_CLUSTERS, DIM), 2e-3), max_iter=1, tol=0.0)
            t_e = (time.perf_counter() - t0) / len(sub)
            out["cpu_baseline"] = {"value": round(1.0 / (10 * t_l + 5 * t_e), 1), "unit": "descriptors/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "NumPy restatement (BLAS threads as configured): 2 Lloyd + 1 EM iteration on 100000 descriptors, "
                                             "scaled to 10 + 5 iterations; seeding and PCA not included"}
    else:  # fp16sim
        L, k = K_CLUSTERS * DIM, 10
        # VLAD-like rows: 256 unit-norm 128-d blocks, ~35 % of them empty
        enc = torch.randn((N, K_CLUSTERS, DIM), generator=g, device=dev)
        enc = enc / enc.norm(dim=2, keepdim=True)
        enc = (enc * (torch.rand((N, K_CLUSTERS, 1), generator=g, device=dev) > 0.35)).reshape(N, L).contiguous()
        e16 = torch.empty((N, L), dtype=torch.float16, device=dev)
        inv = torch.empty((N,), dtype=torch.float32, device=dev)
        idx = torch.empty((N, k), dtype=torch.int64, device=dev)
        val = torch.empty((N, k), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.f32_to_f16_dev(enc.data_ptr(), N * L, e16.data_ptr())
        ctx.row_inv_norms_dev(enc.data_ptr(), N, L, inv.data_ptr())
        ctx.sync()
        del enc
        dt, st = timed(lambda: ctx.cosine_topk_f16_dev(e16.data_ptr(), N, e16.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(),
                                                       k, 0, False, idx.data_ptr(), val.data_ptr()))
        assert np.array_equal(idx[:, 0].cpu().numpy(), np.arange(N))
        flop = 2.0 * N * N * L
        out.update({"metric": "query rows/sec, N x N cosine on fp16 encodings + top-10 (BASELINE configs[4] on one GPU)",
                    "value": round(N / dt, 1), "unit": "rows/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "f16",
                    "scaling": "strong", "stages_ms_per_step": st,
                    "config": {"workload": f"{N} x {N} cosine, L = {L} fp16, fp32 accumulate, top-{k}"},
                    "roofline": {"kernel": "gemm_mfma_kernel<256,256,f16>", "bound": "mfma",
                                 "achieved": round(flop / (st["cosine_gemm"] * 1e-3) / 1e12, 1), "peak": 2500.0,
                                 "unit": "TFLOP/s (algorithmic 2*N*N*L; panels on the diagonal run symmetric)",
                                 "frac": round(flop / (st["cosine_gemm"] * 1e-3) / 1e12 / 2500.0, 4), "traffic": None}})
    print(json.dumps(out))
    ctx.close()


def main():
    args = parse()
    if args.cpu_baseline_only:
        return cpu_baseline_child(args.images)
    if args.workload != "config2":
        return side_workload(args)
    cpu = None
    if not args.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("PVS_BENCH_FORCE_DIST") != "1":
        cpu = cpu_baseline_subprocess(args.images)      # before anything touches the GPU: the child forks worker processes
    import torch
    import torch.distributed as dist
    import pvsim
    from pvsim.engine import DESC_F32, DESC_U8_ROOTSIFT

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # PVS_BENCH_BACKEND=gloo is a REHEARSAL mode for a box with fewer GPUs than ranks: ranks share the visible devices and
    # the collectives go through host copies.  It exercises the whole multi-rank code path except the RCCL transport; its
    # numbers mean nothing.  The measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("PVS_BENCH_BACKEND", "nccl")
    # PVS_BENCH_FORCE_DIST=1 (started through torch.distributed.run with ONE rank): the multi-rank code path -- process group, one
    # stream shared with RCCL, asynchronous exchange, block-pair retrieval, list all-to-all -- on a single GPU; a self-check of
    # that path against the plain single-GPU retrieval, not a measurement
    forced = os.environ.get("PVS_BENCH_FORCE_DIST") == "1" and world == 1
    multi = world > 1 or forced
    if backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # PVS_BENCH_EXCHANGE=neighbours (experimental, default allgather): every rank receives only the blocks its share of the
    # block-pair scheme reads -- (P-1)//2 full blocks plus, for even P, a full or half partner block -- by batched
    # point-to-point transfers instead of the all-gather (-44 % bytes at P = 8)
    exchange_mode = os.environ.get("PVS_BENCH_EXCHANGE", "allgather")
    overlap = os.environ.get("PVS_BENCH_OVERLAP", "1") != "0"      # exchange overlapped with the (r, r) block (RCCL only)

    def coll(fn, out_t, in_t):
        """out_t <- collective(in_t); through host copies in the gloo rehearsal mode"""
        if backend == "gloo":
            o = torch.empty(out_t.shape, dtype=out_t.dtype)
            fn(o, in_t.cpu())
            out_t.copy_(o)
        else:
            fn(out_t, in_t)

    # N > 1: ONE stream for the engine, torch and (through torch's stream semantics) RCCL -- the step then needs no host
    # synchronisation between encode, exchange and retrieval, and the CPU can run ahead of the many small launches
    one_stream = multi
    if one_stream:
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
        ctx = pvsim.Context(local, stream=side.cuda_stream)
    else:
        ctx = pvsim.Context(local)
    tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"), allow_pickle=False)
    cb = ctx.codebook(tables["centroids"])

    # ---- corpus, sharded by image: rank r owns images [lo, hi)
    N = args.images
    raw_all_counts_seed = 1235
    per = (N + world - 1) // world
    lo, hi = min(N, rank * per), min(N, (rank + 1) * per)
    n_loc = hi - lo
    # every rank generates only its own shard (seeded per rank so shards differ)
    raw, offsets = make_corpus(n_loc, raw_all_counts_seed + 7919 * rank, dev)
    total_desc = int(offsets[-1])
    kind = DESC_U8_ROOTSIFT if args.desc == "u8" else DESC_F32
    desc = raw if args.desc == "u8" else rootsift_torch(raw)
    d_off = torch.from_numpy(offsets).to(dev)
    L = K_CLUSTERS * DIM
    enc_loc = torch.empty((per, L), dtype=torch.float32, device=dev)       # padded to the common block size
    inv_loc = torch.ones((per,), dtype=torch.float32, device=dev)
    if n_loc < per:
        enc_loc[n_loc:].zero_()
    if multi:
        enc_all = torch.empty((world * per, L), dtype=torch.float32, device=dev)
        inv_all = torch.empty((world * per,), dtype=torch.float32, device=dev)
    else:
        enc_all, inv_all = enc_loc, inv_loc
    idx = torch.empty((max(n_loc, 1), TOPK), dtype=torch.int64, device=dev)
    val = torch.empty((max(n_loc, 1), TOPK), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    from pvsim import distributed as pd
    score_block = pd.device_score_block(ctx)
    ops = pd.DeviceOps(ctx, same_stream=one_stream)

    def a2a(out_t, in_t):
        coll(dist.all_to_all_single, out_t, in_t)
        if not one_stream:
            torch.cuda.current_stream().synchronize()

    def new_tensor(shape, dtype, fill):
        t_ = torch.full(shape, fill, dtype=dtype, device=dev)
        if not one_stream:
            torch.cuda.current_stream().synchronize()      # visible to the context's stream
        return t_

    def rows_of(r):
        return max(0, min(N, (r + 1) * per) - r * per)

    def exchange_neighbours():
        P, B, h = world, per, (world - 1) // 2
        sends, recvs = [], []                                   # (peer, row0, row1) of MY block / of the peer's block
        for j in range(1, h + 1):
            sends.append(((rank - j) % P, 0, n_loc))
            recvs.append(((rank + j) % P, 0, rows_of((rank + j) % P)))
        if P % 2 == 0:
            pr = (rank + P // 2) % P
            a, b = min(rank, pr), max(rank, pr)
            hb = min(rows_of(b), (B + 1) // 2)
            if rank == a:                                       # a scores Q_a x DB_b[0:hb]; b scores Q_a x DB_b[hb:]
                sends.append((b, 0, n_loc))
                recvs.append((b, 0, hb))
            else:
                sends.append((a, 0, hb))
                recvs.append((a, 0, rows_of(a)))
        enc_all[rank * B:rank * B + n_loc].copy_(enc_loc[:n_loc])
        ops, stage = [], []
        for peer, r0, r1 in sends:
            if r1 > r0:
                t_ = enc_loc[r0:r1]
                if backend == "gloo":
                    t_ = t_.cpu()
                ops.append(dist.P2POp(dist.isend, t_, peer))
        for peer, r0, r1 in recvs:
            if r1 > r0:
                dst = enc_all[peer * B + r0:peer * B + r1]
                if backend == "gloo":
                    buf = torch.empty(dst.shape, dtype=dst.dtype)
                    stage.append((dst, buf))
                    dst = buf
                ops.append(dist.P2POp(dist.irecv, dst, peer))
        works = dist.batch_isend_irecv(ops) if ops else []
        if backend == "gloo" or not overlap:
            for w_ in works:
                w_.wait()
            for dst, buf in stage:
                dst.copy_(buf)
            return []
        return [w_.wait for w_ in works]

    def exchange_begin():
        """Start the exchange of the encoded blocks; returns the waiters to call before another rank's rows are read.  With RCCL
        the collectives are asynchronous (they run on the process group's own stream after this stream's encode, and a waiter
        only makes THIS stream wait for them), so the (r, r) block is scored while the blocks travel.  The gloo rehearsal
        stages through the host and is synchronous."""
        sync_mode = backend == "gloo" or not overlap
        if exchange_mode == "neighbours":
            ws = exchange_neighbours()
        elif sync_mode:
            coll(dist.all_gather_into_tensor, enc_all, enc_loc)
            ws = []
        else:
            ws = [dist.all_gather_into_tensor(enc_all, enc_loc, async_op=True).wait]
        if sync_mode:
            coll(dist.all_gather_into_tensor, inv_all, inv_loc)
        else:
            ws.append(dist.all_gather_into_tensor(inv_all, inv_loc, async_op=True).wait)
        return ws

    def step():
        ctx.vlad_encode_dev(cb, desc.data_ptr(), kind, d_off.data_ptr(), n_loc, total_desc, enc_loc.data_ptr(),
                            d_inv_norm=inv_loc.data_ptr())
        waiters = []
        if multi:
            if not one_stream:
                ctx.sync()                               # encode (ctx stream) -> collective (torch stream)
            waiters = exchange_begin()

        def exchanged():
            for w_ in waiters:
                w_()
            if not one_stream:
                torch.cuda.current_stream().synchronize()

        if not multi and filtered[0]:
            filt_stats[0] = ctx.cosine_topk_filtered_dev(enc_loc.data_ptr(), n_loc, enc_loc.data_ptr(), n_loc, L, inv_loc.data_ptr(),
                                                         inv_loc.data_ptr(), TOPK, idx.data_ptr(), val.data_ptr())
        elif not multi:
            pd.retrieve_sharded(enc_loc, inv_loc, enc_all, inv_all, N, rank, world, TOPK, score_block, idx, val)
        elif args.retrieval == "filtered":
            exchanged()
            # opt-in: this rank's queries against the gathered corpus through the prefilter + exact re-scoring (no list
            # exchange needed; the first N gathered rows are the real ones, padding only follows the last block)
            ctx.cosine_topk_filtered_dev(enc_loc.data_ptr(), n_loc, enc_all.data_ptr(), N, L, inv_loc.data_ptr(), inv_all.data_ptr(),
                                         TOPK, idx.data_ptr(), val.data_ptr())
        else:
            # every block pair is scored once (dual-store GEMM), k-candidate lists exchanged, merged
            # the (r, r) block reads the local copy, so it runs while the other blocks are still arriving
            i_, v_ = pd.retrieve_symmetric(enc_all, inv_all, N, rank, world, TOPK, ops, a2a, new_tensor,
                                           own=(enc_loc, inv_loc), before_cross=exchanged)
            idx[:n_loc].copy_(i_)
            val[:n_loc].copy_(v_)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps():
        for _ in range(args.warmup):
            step()
        barrier()
        ctx.timers_enable(True)
        ctx.timers_reset()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_ = time.perf_counter() - t0_
        tm_ = ctx.timers()
        ctx.timers_enable(False)
        return dt_, tm_

    filtered = [args.retrieval == "filtered" and not multi]   # (N > 1 reads args.retrieval directly in step())
    filt_stats = [None]
    other = None
    if not multi:
        # the variant that is NOT the headline of this run is timed first, its lists kept for the bit-for-bit comparison
        filtered[0] = not filtered[0]
        o_dt, o_tm = timed_steps()
        other = {"dt": o_dt, "timers": o_tm, "idx": idx.clone(), "val": val.clone(), "was_filtered": filtered[0], "stats": filt_stats[0]}
        filtered[0] = not filtered[0]
    dt, timers = timed_steps()
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = N / (dt / args.steps)

    # ---- sanity inside the bench: self-retrieval must return the image itself first
    got = idx[:n_loc, 0].cpu().numpy()
    assert np.array_equal(got, np.arange(lo, hi)), "self-retrieval failed: top-1 is not the query image"
    if multi and (backend == "gloo" or forced):
        # rehearsal: the block-pair scheme must give exactly what one GPU gives for this rank's queries
        ri = torch.empty((n_loc, TOPK), dtype=torch.int64, device=dev)
        rv = torch.empty((n_loc, TOPK), dtype=torch.float32, device=dev)
        inv_ref = inv_all.clone()
        inv_ref[N:] = float("nan")                     # padding rows of the last block never rank
        if exchange_mode == "neighbours":              # enc_all holds only this rank's partner blocks: gather everything for the check
            coll(dist.all_gather_into_tensor, enc_all, enc_loc)
        torch.cuda.synchronize()
        ctx.cosine_topk_dev(enc_loc.data_ptr(), n_loc, enc_all.data_ptr(), min(world * per, enc_all.shape[0]), L, inv_loc.data_ptr(),
                            inv_ref.data_ptr(), TOPK, 0, False, ri.data_ptr(), rv.data_ptr())
        ctx.sync()
        assert torch.equal(ri, idx[:n_loc]) and torch.equal(rv.view(torch.int32), val[:n_loc].view(torch.int32)), \
            f"rank {rank}: multi-rank retrieval differs from the single-GPU ranking"
        print(f"[rehearsal] rank {rank}: {n_loc} queries identical to the single-GPU ranking", file=sys.stderr)

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (the cosine GEMM): EXECUTED flop / measured launch duration.
    # One launch = local queries x one rank block.  When the block is the query block itself (self-similarity)
    # the kernel computes only the upper triangle of 128x128 tiles and mirrors the rest, so the executed flop is
    # ~half of the algorithmic 2*N*M*L of SURVEY.md section 8(d); `achieved` counts executed flop only (never above peak).
    gemm_ms, gemm_n = timers["cosine_gemm"]
    t128 = (n_loc + 127) // 128
    if not multi:
        alg_flop = 2.0 * N * N * L
        flop_per_launch = 2.0 * 128 * 128 * L * (t128 * (t128 + 1) // 2)      # upper-triangle tiles
    else:
        alg_flop = 2.0 * n_loc * N * L
        # symmetric pair scheme: own block (upper triangle) + (P-1)/2 cross blocks, over several launches per step
        launches_per_step = max(gemm_n / max(args.steps, 1), 1.0)
        flop_per_launch = 2.0 * 128 * 128 * L * (t128 * (t128 + 1) / 2 + (world - 1) / 2.0 * t128 * t128) / launches_per_step
    gemm_avg_ms = gemm_ms / max(gemm_n, 1)
    achieved = flop_per_launch / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0
    # the exact f32 GEMM only (the same process also runs the fp16 prefilter GEMM of the filtered variant)
    traffic, traffic_src = pmc_traffic("pvs::gemm_mfma_kernel", keep=lambda nm: not _gemm_is_f16(nm))
    if multi or N != 8189 or filtered[0]:
        traffic, traffic_src = None, None             # the committed counters are for the default 1-GPU workload
    stages = {k: {"ms_total": round(v[0], 3), "launches": int(v[1]),
                  "ms_avg": round(v[0] / v[1], 4) if v[1] else None} for k, v in timers.items() if v[1]}
    enc_ms = (timers["assign"][0] + timers["aggregate"][0]) / args.steps
    desc_bytes = total_desc * DIM * (1 if args.desc == "u8" else 4)
    enc_bytes = desc_bytes + n_loc * L * 4
    out = {
        "metric": "images/sec encoded + top-k retrieved, VLAD K256 RootSIFT",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        **({"exchange": exchange_mode, "exchange_overlaps_own_block": bool(overlap and backend != "gloo")} if multi else {}),
        **({"backend": "ONE-rank RCCL self-check of the multi-rank path: not a measurement"} if forced else {}),
        **({"backend": "gloo REHEARSAL (ranks share GPUs, host-staged collectives): not a measurement"} if backend == "gloo" else {}),
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "retrieval": args.retrieval,
        "config": {"workload": f"configs[1]: {N} images x ragged SIFT-like descriptors (mean {total_desc / max(n_loc, 1):.0f}/image),"
                               f" D=128, VLAD K=256 encode + {N}x{N} cosine + top-{TOPK}",
                   "images": N, "descriptors_rank0": total_desc, "descriptor_rows": args.desc,
                   "K": K_CLUSTERS, "D": DIM, "topk": TOPK, "parallelism": f"image-sharded x{world}"},
        "roofline": {"kernel": ("gemm_mfma_kernel<128,128,f16, chains of 1024 k> (prefilter GEMM of the filtered retrieval)" if filtered[0] else
                                "gemm_mfma_kernel<128,128,f32> (cosine GEMM: main + split-K tail)"), "bound": "mfma", "achieved": round(achieved, 2),
                     "peak": 2500.0 if filtered[0] else FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / (2500.0 if filtered[0] else FP32_MFMA_PEAK_TFLOPS), 4),
                     "traffic": traffic, "traffic_source": traffic_src, "flop_per_launch": flop_per_launch,
                     "mfma_pipe_busy_frac": None if (multi or N != 8189 or filtered[0]) else
                     pmc_mfma_busy("pvs::gemm_mfma_kernel", keep=lambda nm: not _gemm_is_f16(nm)),
                     "algorithmic_flop_per_launch": alg_flop,
                     "algorithmic_equiv_TFLOPs": round(alg_flop / (gemm_avg_ms * 1e-3) / 1e12, 2) if gemm_n else None, "avg_launch_ms": round(gemm_avg_ms, 4)},
        "stages": stages,
        "encode": {"ms_per_step": round(enc_ms, 3), "images_per_s": round(n_loc / (enc_ms * 1e-3), 1) if enc_ms else None,
                   "algorithmic_GBps": round(enc_bytes / (enc_ms * 1e-3) / 1e9, 1) if enc_ms else None,
                   "assign_TFLOPs": round(2.0 * total_desc * K_CLUSTERS * DIM / (timers["assign"][0] / args.steps * 1e-3) / 1e12, 2)
                   if timers["assign"][0] else None},
        "device": ctx.device_name(),
    }

    if other is not None:
        same_lists = bool(torch.equal(other["idx"], idx)) and bool(torch.equal(other["val"].view(torch.int32), val.view(torch.int32)))
        f_dt, f_tm, f_st = (other["dt"], other["timers"], other["stats"]) if other["was_filtered"] else (dt, timers, filt_stats[0])
        e_dt = dt if other["was_filtered"] else other["dt"]
        out["retrieval"] = args.retrieval
        out["filtered_retrieval"] = {
            "what": "same top-k lists through pvs_cosine_topk_filtered_dev: fp16 MFMA prefilter with a proven error bound + exact "
                    "fp32 re-scoring of the candidates (bit-identical indices and scores; opt-in, --retrieval filtered)",
            "ms_per_step": round(f_dt / args.steps * 1e3, 3), "images_per_s": round(N / (f_dt / args.steps), 1),
            "exact_ms_per_step": round(e_dt / args.steps * 1e3, 3),
            "lists_bit_identical_to_exact": same_lists,
            "stages_ms": {k: round(v[0] / args.steps, 4) for k, v in f_tm.items() if v[1]},
            "candidates_per_query": round(f_st["candidates"] / max(n_loc, 1), 2) if f_st else None,
            "queries_redone_exact": f_st["redone_exact"] if f_st else None}
        assert same_lists, "filtered retrieval differs from the exact path"

    if args.pcie and not multi:
        h_desc = desc.cpu().numpy()
        t1 = time.perf_counter()
        v = ctx.vlad_encode(cb, h_desc, offsets, kind)
        ctx.cosine_topk(v, v, TOPK)
        out["pcie_inclusive_images_per_s"] = round(N / (time.perf_counter() - t1), 1)

    if cpu is not None and not multi:
        out["cpu_baseline"] = cpu

    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the NumPy restatement of the reference's per-image / per-query procedure
# (oracle/pvsim_oracle.py: KMeans.predict as an fp32 GEMM + argmin -> residual sums in descriptor order -> normalise;
# per query: cosine against the WHOLE database, re-normalised every time as sklearn does, -> full argsort -> first k),
# timed on this host's cores in two modes:
#   A  one process, library-default threading -- what a pyvisim user gets
#   B  one worker process per core, BLAS / OpenMP threads pinned to 1 -- best effort; the denominator of any GPU / CPU ratio
# It runs in a CHILD process started before this process touches the GPU (so it may fork workers freely) on a bounded sample:
# ragged SIFT-like images from the same generator, a 2048-row database; the per-query cost is scaled to the full database
# size (it is linear in the rows scanned).
CPU_DB_ROWS = 2048


def _cpu_env_threads(n):
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[k] = str(n)


_W = {}


def _w_init(centroids, one_thread=True):
    if one_thread:
        try:
            from threadpoolctl import threadpool_limits
            _W["lim"] = threadpool_limits(1)
        except Exception:
            pass
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import pvsim_oracle as orc
    from pvsim import synth
    _W.update(orc=orc, synth=synth, C=centroids)


def _w_encode(job):
    seed, n_img = job
    orc, synth = _W["orc"], _W["synth"]
    counts = synth.ragged_counts(n_img, seed)
    rng = np.random.default_rng(seed)
    raws = [synth.sift_like(int(c), rng) for c in counts]
    t0 = time.perf_counter()
    v = np.vstack([orc.vlad_encode_one(orc.rootsift(r), _W["C"]) for r in raws])
    return v, time.perf_counter() - t0, int(sum(counts))


def _w_query(job):
    lo, hi = job
    orc, db = _W["orc"], _W["db"]
    t0 = time.perf_counter()
    for i in range(lo, hi):
        orc.retrieve_top_k(db[i], db, TOPK)
    return time.perf_counter() - t0


def cpu_baseline_child(n_full):
    """Runs in the child process: prints one JSON object."""
    import multiprocessing as mp
    import platform
    tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"), allow_pickle=False)
    C = np.ascontiguousarray(tables["centroids"], dtype=np.float32)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    P = max(1, min(cores, 64))
    # ---- mode B first (it also builds the database): P single-thread workers
    ctx = mp.get_context("fork")
    per = max(1, CPU_DB_ROWS // P)
    jobs = [(9000 + i, per) for i in range(P)]
    t0 = time.perf_counter()
    with ctx.Pool(P, initializer=_w_init, initargs=(C,)) as pool:
        parts = pool.map(_w_encode, jobs)
    wall_b_enc = time.perf_counter() - t0
    db = np.ascontiguousarray(np.vstack([p[0] for p in parts]))
    n_b = db.shape[0]
    mean_desc = sum(p[2] for p in parts) / n_b
    # wall time includes the worker start-up; the workers' own clocks give the steady-state rate
    enc_b = max(p[1] for p in parts) / per / P * 1.0      # seconds per image at P-way throughput
    _W["db"] = db
    q_per = 2
    qjobs = [(i * q_per, (i + 1) * q_per) for i in range(P)]
    with ctx.Pool(P, initializer=_w_init, initargs=(C,)) as pool:
        t0 = time.perf_counter()
        qt = pool.map(_w_query, qjobs)
        wall_b_q = time.perf_counter() - t0
    ret_b = max(qt) / q_per / P * (n_full / n_b)          # seconds per query against the FULL database, P-way throughput
    # ---- mode A: this process, default threads
    _w_init(C, one_thread=False)
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: d.get(k) for k in ("internal_api", "version", "num_threads")} for d in threadpool_info()]
    except Exception:
        pools = None
    n_a = 128
    _, t_a, _ = _w_encode((7777, n_a))
    enc_a = t_a / n_a
    nq_a = 8
    t0 = time.perf_counter()
    for i in range(nq_a):
        _W["orc"].retrieve_top_k(db[i], db, TOPK)
    ret_a = (time.perf_counter() - t0) / nq_a * (n_full / n_b)
    out = {"value": round(1.0 / (enc_b + ret_b), 2), "unit": "images/s", "cores": P, "kind": "port",
           "sample": f"NumPy restatement of the reference's procedure (oracle/pvsim_oracle.py). Mode B (value): {P} worker processes, 1 BLAS "
                     f"thread each: {n_b} ragged images encoded (mean {mean_desc:.0f} descriptors), {P * q_per} queries ranked against the "
                     f"{n_b}-row database; per-query cost scaled x{n_full / n_b:.2f} to the {n_full}-row database. Mode A: one process, "
                     f"library-default threads: {n_a} images encoded, {nq_a} queries",
           "mode_B": {"images_per_s": round(1.0 / (enc_b + ret_b), 2), "encode_images_per_s": round(1.0 / enc_b, 1),
                      "retrieve_queries_per_s": round(1.0 / ret_b, 2), "workers": P, "threads_per_worker": 1,
                      "encode_wall_s_incl_startup": round(wall_b_enc, 2), "query_wall_s": round(wall_b_q, 2)},
           "mode_A": {"images_per_s": round(1.0 / (enc_a + ret_a), 2), "encode_images_per_s": round(1.0 / enc_a, 1),
                      "retrieve_queries_per_s": round(1.0 / ret_a, 2), "threadpools": pools},
           "host": {"cpus_available": cores, "os_cpu_count": os.cpu_count(), "machine": platform.processor() or platform.machine(),
                    "numpy": np.__version__}}
    try:
        with open("/proc/cpuinfo") as f:
            names = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        out["host"]["cpu_model"] = names[0] if names else None
    except OSError:
        pass
    print(json.dumps(out))


def cpu_baseline_subprocess(n_full):
    """Start the CPU baseline as a child process.  Must be called BEFORE this process initialises the GPU."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--images", str(n_full)],
                       capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return {"error": r.stderr[-800:]}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": "no output"}


if __name__ == "__main__":
    main()
