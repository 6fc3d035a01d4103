#!/usr/bin/env python3
"""bench.py -- images/sec encoded + top-k retrieved, VLAD K=256 RootSIFT (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over the whole synthetic corpus, inputs already resident in HBM:
  encode  : descriptors -> KMeans assignment -> VLAD residual aggregation -> power + intra L2 norm
  exchange: (N > 1) RCCL all-gather of the per-GPU encoding blocks (pvs_comm_* behind the C-ABI)
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
    ap.add_argument("--images", type=int, default=None, help="default: 8189 (configs[1]/[2]); 1,000,000 for --workload corpus1m")
    ap.add_argument("--desc", choices=["f32", "u8"], default="f32",
                    help="descriptor rows in HBM: fp32 RootSIFT (default) or raw uint8 SIFT with fused RootSIFT")
    ap.add_argument("--queries", type=int, default=0, help="query rows per rank (0 = every local image: all-vs-all)")
    ap.add_argument("--total-queries", type=int, default=0,
                    help="query rows over ALL ranks (strong scaling: each rank ranks total/N of them); overrides --queries")
    ap.add_argument("--prefilter-products", type=int, choices=[1, 2, 3], default=3,
                    help="fp16 products per (row, cluster) of the assignment prefilter: 3 = the product path; 2 / 1 = measurement "
                         "variants with a wider margin (more rows left to the exact kernel; same labels)")
    ap.add_argument("--prefilter-16x16", action="store_true",
                    help="assignment prefilter on v_mfma_f32_16x16x32_f16 (PVS_OPT_ASSIGN_PREFILTER = 4: measurement variant, 8-10 %% slower); "
                         "same labels")
    ap.add_argument("--fisher-scale", type=int, choices=[0, 1, 2], default=0,
                    help="--workload fisher: PVS_OPT_FISHER_SCALE (0 / 1 = norm division as a second pass, the default; 2 = inside the moments kernel)")
    ap.add_argument("--fused", action="store_true", help="encode with the one-read fused kernel (PVS_OPT_VLAD_PATH = 3) instead of assign + aggregate")
    ap.add_argument("--workload", choices=["config2", "fisher", "vlad512", "fp16sim", "learn", "corpus1m", "serve"], default=None,
                    help="default: config2 (BASELINE configs[1], the headline) with --gpus 1; corpus1m (configs[3]/[4]: 1M images "
                         "sharded over the ranks, fp16 exchange + retrieval, 65536 queries in all) with --gpus N > 1.  Side workloads (single GPU, same JSON shape, not the "
                         "headline): fisher = BASELINE configs[2] (Fisher D=512 K=256 n=196), vlad512 = the per-GPU share of "
                         "configs[3] (n=512 descriptors per image, encode only), fp16sim = configs[4] scaled to one GPU "
                         "(N x N cosine on fp16 encodings + top-10), serve = ONE query image at a time through the class API "
                         "(eval.retrieve_top_k_similar: encode + top-5) against the 8189-image index, resident (pvsim.index.DeviceIndex) and "
                         "as the reference's plain dict")
    ap.add_argument("--retrieval", choices=["exact", "filtered", "f16", "f64"], default=None,
                    help="f64 (--workload fisher only) = float64 encodings scored and ranked in float64 on the f64 matrix pipe, the "
                         "reference's dtype for Fisher vectors.  exact = f32 MFMA GEMM over all pairs (the headline). filtered = the same top-k lists, bit for bit, "
                         "through the fp16 prefilter + exact re-scoring (pvs_cosine_topk_filtered_dev); single GPU only. The "
                         "default run also times the filtered variant and reports it under 'filtered_retrieval'.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of cpu_baseline_subprocess()
    ap.add_argument("--pcie", action="store_true", help="also time one step with host-resident inputs (H2D included)")
    args = ap.parse_args()
    # defaults that depend on the GPU count: the driver types `bench.py --gpus N --steps K --warmup W` and nothing else
    args.workload_defaulted = args.workload is None
    if args.workload is None:
        args.workload = "config2" if args.gpus <= 1 else "corpus1m"
    if args.images is None:
        args.images = 1000000 if args.workload == "corpus1m" else 8189
    if args.retrieval is None:
        args.retrieval = "f16" if (args.workload == "corpus1m" and args.workload_defaulted) else "exact"
    if args.workload == "corpus1m" and args.workload_defaulted and args.queries == 0 and args.total_queries == 0:
        args.total_queries = 65536
    if args.retrieval == "f64" and args.workload != "fisher":
        ap.error("--retrieval f64 belongs to --workload fisher")
    return args


def self_launch(args):
    """`python3 bench.py --gpus N` typed without a launcher: start the N ranks as FRESH processes (this one has not touched the
    GPU and never will), one per GPU, through torch.distributed.run on 127.0.0.1; rank 0's JSON line goes to our stdout as it is
    printed, the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


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
        if args.fisher_scale:
            from pvsim import _ffi
            ctx.set_option(_ffi.OPT_FISHER_SCALE, args.fisher_scale)
            out["fisher_scale_option"] = args.fisher_scale
        z = torch.randint(0, K, (N * n,), generator=g, device=dev)
        desc = (torch.from_numpy(means.astype(np.float32)).to(dev)[z]
                + torch.from_numpy(np.sqrt(cov).astype(np.float32)).to(dev)[z] * torch.randn((N * n, D), generator=g, device=dev))
        off = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * n
        L = K + 2 * K * D
        f64 = args.retrieval == "f64"      # the reference's dtype: float64 encodings, float64 scores, float64 ranking
        enc = torch.empty((N, L), dtype=torch.float64 if f64 else torch.float32, device=dev)
        inv = torch.empty((N,), dtype=torch.float64 if f64 else torch.float32, device=dev)
        idx = torch.empty((N, TOPK), dtype=torch.int64, device=dev)
        val = torch.empty((N, TOPK), dtype=torch.float64 if f64 else torch.float32, device=dev)
        torch.cuda.synchronize()
        chunk = 32768

        def step():
            for s0 in range(0, N, chunk):
                e0 = min(N, s0 + chunk)
                ctx.fisher_encode_dev(gm, desc[s0 * n:].data_ptr(), DESC_F32, (off[s0:e0 + 1] - off[s0]).contiguous().data_ptr(),
                                      e0 - s0, (e0 - s0) * n, enc[s0:].data_ptr(), 1 if f64 else 0)
            if f64:
                ctx.row_inv_norms_f64_dev(enc.data_ptr(), N, L, inv.data_ptr())
                ctx.cosine_topk_f64_dev(enc.data_ptr(), N, enc.data_ptr(), N, L, inv.data_ptr(), inv.data_ptr(), TOPK,
                                        idx.data_ptr(), val.data_ptr())
                return
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
        t128 = (N + 127) // 128
        exec_flop = 2.0 * 128 * 128 * L * (t128 * (t128 + 1) // 2)     # the symmetric kernels run the upper triangle of tiles
        gemm_ms = st.get("cosine_gemm", 0)
        peak = 78.6 if f64 else FP32_MFMA_PEAK_TFLOPS
        out.update({"metric": "images/sec encoded + top-k retrieved, Fisher K256 D512 n196 (BASELINE configs[2])",
                    "value": round(N / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3),
                    "dtype": "f64" if f64 else "f32 (scores and stored encodings; the encode arithmetic is f64)",
                    "scaling": "strong", "stages_ms_per_step": st,
                    "config": {"workload": f"{N} images x {n} x {D}-D descriptors, diag GMM K={K}: Fisher encode (fp64 arithmetic, "
                                           f"{'fp64' if f64 else 'fp32'} stored) + {N}x{N} cosine in {'float64' if f64 else 'float32'} + top-{TOPK}",
                               "K": K, "D": D},
                    "roofline": {"kernel": "gemm_f64_kernel (v_mfma_f64_16x16x4_f64, symmetric)" if f64 else "gemm_mfma_kernel<128,128,f32> (symmetric)",
                                 "bound": "mfma", "achieved": round(exec_flop / (gemm_ms * 1e-3) / 1e12, 2) if gemm_ms else None,
                                 "peak": peak, "unit": "TFLOP/s (executed: upper triangle of 128x128 tiles)",
                                 "frac": round(exec_flop / (gemm_ms * 1e-3) / 1e12 / peak, 4) if gemm_ms else None, "traffic": None,
                                 "algorithmic_equiv_TFLOPs": round(2.0 * N * N * L / (gemm_ms * 1e-3) / 1e12, 2) if gemm_ms else None},
                    "encode_GFLOPs_fp64": round(8.0 * n * K * D * N / (enc_ms * 1e-3) / 1e9, 1) if enc_ms else None})
    elif args.workload == "vlad512":
        n = 512
        cb = ctx.codebook(tables["centroids"])
        if args.fused:
            from pvsim import _ffi
            ctx.set_option(_ffi.OPT_VLAD_PATH, _ffi.VLAD_PATH_FUSED)
        if args.prefilter_products != 3:
            from pvsim import _ffi
            ctx.set_option(_ffi.OPT_ASSIGN_PREFILTER, {2: 2, 1: 3}[args.prefilter_products])
        elif args.prefilter_16x16:
            from pvsim import _ffi
            ctx.set_option(_ffi.OPT_ASSIGN_PREFILTER, 4)
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
                    "config": {"workload": f"{N} images x 512 SIFT-like descriptors ({args.desc}), VLAD K=256 encode only",
                               "encode_path": "fused one-read kernel" if args.fused else "assign + aggregate",
                               "prefilter_products": args.prefilter_products},
                    # the whole encode (assign + aggregate) against HBM: descriptors read twice (K1, K2) is what the kernels do,
                    # the algorithmic bytes count them once (393,216 B / image from f32 rows)
                    "roofline": {"kernel": "vlad_fused_kernel (one read)" if args.fused else "assign16_kernel + assign_kernel (near ties) + vlad_aggregate_kernel", "bound": "hbm",
                                 "achieved": round(byt / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(byt / dt / 1e9 / 8000.0, 4), "traffic": None,
                                 "assign_executed_f16_TFLOPs": None if args.fused else round(3 * 2.0 * N * n * K_CLUSTERS * DIM / (st["assign"] * 1e-3) / 1e12, 1)},
                    "encode_algorithmic_GBps": round(byt / dt / 1e9, 1)})
    elif args.workload == "serve":
        # Latency, not throughput: pyvisim's interactive use (eval.py:13-46) -- one uploaded image, encode, rank against the index,
        # k paths back.  Host descriptors in, host list out: PCIe and every launch are inside the timed call.
        from pvsim import eval as pe, synth
        from pvsim.encoders import VLADEncoder
        from pvsim.features import Lambda
        from pvsim.index import DeviceIndex
        from pvsim.models import KMeansModel
        raw, offsets = make_corpus(N, 1235, dev)
        d_off = torch.from_numpy(offsets).to(dev)
        desc = rootsift_torch(raw)
        cb = ctx.codebook(tables["centroids"])
        enc = torch.empty((N, K_CLUSTERS * DIM), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()                              # torch filled `desc` on its own stream
        ctx.vlad_encode_dev(cb, desc.data_ptr(), DESC_F32, d_off.data_ptr(), N, int(offsets[-1]), enc.data_ptr())
        ctx.sync()
        db = enc.cpu().numpy()
        paths = [f"img_{i:05d}.jpg" for i in range(N)]
        enc_map = dict(zip(paths, db))
        rng = np.random.default_rng(4)
        q_ids = rng.integers(0, N, size=64)
        q_raw = [raw[offsets[i]:offsets[i + 1]].cpu().numpy() for i in q_ids]        # the "images": raw SIFT-like uint8 descriptors
        encoder = VLADEncoder(Lambda(lambda im: synth.rootsift(im.astype(np.float32)), DIM), kmeans_model=KMeansModel(tables["centroids"]),
                              context=ctx)
        index = DeviceIndex(enc_map, ctx)

        def run(dataset, reps):
            lat = []
            for r in range(reps):
                qi = r % len(q_raw)
                t0 = time.perf_counter()
                got = pe.retrieve_top_k_similar([q_raw[qi]], dataset, encoder, k=TOPK)
                lat.append(time.perf_counter() - t0)
                assert got[0][0] == paths[q_ids[qi]], f"the query image {paths[q_ids[qi]]} is not its own nearest neighbour: {got}"
            return np.array(lat) * 1e3
        run(index, max(args.warmup, 1) * 8)
        lat = run(index, max(args.steps, 1) * 64)
        lat_dict = run(enc_map, 4)
        a_ = pe.retrieve_top_k_similar([q_raw[0]], enc_map, encoder, k=TOPK)
        b_ = pe.retrieve_top_k_similar([q_raw[0]], index, encoder, k=TOPK)
        assert a_ == b_, "resident index and plain dict disagree"
        out.update({"metric": "queries/sec, one image at a time: RootSIFT + VLAD K256 encode + top-5 against a resident 8189-image index (class API, host in / host out)",
                    "value": round(1e3 / float(np.median(lat)), 1), "unit": "queries/s", "ms_per_step": round(float(np.median(lat)), 4),
                    "higher_is_better": True, "dtype": "f32", "scaling": "strong",
                    "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 4), "p90": round(float(np.percentile(lat, 90)), 4),
                                   "p99": round(float(np.percentile(lat, 99)), 4), "min": round(float(lat.min()), 4), "calls": int(lat.size)},
                    "plain_dict_latency_ms": {"p50": round(float(np.median(lat_dict)), 2), "calls": int(lat_dict.size),
                                              "what": "the reference's calling convention: the (N, L) matrix is rebuilt from the dict and uploaded on every call"},
                    "same_result_as_plain_dict": True,
                    "config": {"workload": f"{N}-image VLAD index ({db.nbytes / 1e9:.2f} GB float32) resident on the GPU; queries of {int(np.mean([len(q) for q in q_raw]))} descriptors on average",
                               "K": K_CLUSTERS, "D": DIM, "topk": TOPK},
                    "roofline": {"kernel": "latency-bound (one 1 x N similarity row: 1.07 GB read per query)", "bound": "hbm",
                                 "achieved": round(db.nbytes / (float(np.median(lat)) * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(db.nbytes / (float(np.median(lat)) * 1e-3) / 1e9 / 8000.0, 4), "traffic": None}})
        index.close()
        if not args.no_cpu_baseline:
            # the reference's procedure for ONE uploaded image (eval.py:13-46), restated in NumPy: RootSIFT -> predict -> residual loop ->
            # normalise; rebuild the matrix from the dict; cosine against the whole database (re-normalised, as sklearn does); argsort
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import pvsim_oracle as orc
            cen = np.asarray(tables["centroids"], dtype=np.float32)
            t_cpu = []
            for qi in range(3):
                t0 = time.perf_counter()
                v = orc.vlad_encode([synth.rootsift(q_raw[qi].astype(np.float32))], cen)
                allv = np.array(list(enc_map.values()))
                sc = orc.cosine_similarity(v, allv)[0]
                top = np.argsort(-sc)[:TOPK]
                t_cpu.append(time.perf_counter() - t0)
                assert paths[int(top[0])] == paths[q_ids[qi]]
            out["cpu_baseline"] = {"value": round(1.0 / float(np.median(t_cpu)), 2), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"NumPy restatement of the reference's per-query procedure (oracle/pvsim_oracle.py, BLAS threads as configured): "
                                             f"3 queries against the same {N}-image database, median {float(np.median(t_cpu)) * 1e3:.0f} ms per query"}
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
        res, samples = {}, {}

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
            for k_, v_ in (("seeding_s", t1 - t0), ("lloyd10_s", t2 - t1), ("em5_s", t3 - t2), ("pca_s", t4 - t3)):
                samples.setdefault(k_, []).append(v_)
            res.update(inertia=km.inertia_, lower_bound=gm.lower_bound_)

        dt, st = timed(one)
        res.update({k_: float(np.median(v_[-max(args.steps, 1):])) for k_, v_ in samples.items()})     # medians over the timed steps
        out.update({"metric": "descriptors/sec through k-means++ seeding + 10 Lloyd iterations + 5 EM iterations + PCA fit (K=256, D=128)",
                    "value": round(total / dt, 1), "unit": "descriptors/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "f32 (k-means), f64 (EM, PCA)",
                    "scaling": "strong", "stages_ms_per_step": st, "phases_s": {k: round(v, 4) for k, v in res.items()},
                    "phase_samples_s": {k_: [round(t_, 4) for t_ in v_[-max(args.steps, 1):]] for k_, v_ in samples.items()},
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
                    "roofline": {"kernel": "gemm_f16_8ph_kernel<256x256, v_mfma_f32_16x16x32_f16, 8-phase schedule>", "bound": "mfma",
                                 "achieved": round(flop / (st["cosine_gemm"] * 1e-3) / 1e12, 1), "peak": 2500.0,
                                 "unit": "TFLOP/s (algorithmic 2*N*N*L; panels on the diagonal run symmetric)",
                                 "frac": round(flop / (st["cosine_gemm"] * 1e-3) / 1e12 / 2500.0, 4), "traffic": None}})
    print(json.dumps(out))
    ctx.close()


class GlooStagedComm:
    """REHEARSAL transport for a box with fewer GPUs than ranks: torch.distributed (gloo) with host-staged copies behind the
    call contract of pvsim.distributed.RcclComm.  It exercises the whole multi-rank code path except the RCCL transport; its
    numbers mean nothing.  The measured configuration is always RcclComm, one rank per GPU."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def all_gather(self, send, recv):
        import torch
        o = torch.empty((recv.numel(),), dtype=recv.dtype)          # flat: recv may be (world, rows, L) or (world * rows, L)
        self.dist.all_gather_into_tensor(o, send.contiguous().cpu().reshape(-1))
        recv.view(-1).copy_(o)

    def all_to_all(self, out, inp):
        import torch
        o = torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_to_all_single(o, inp.cpu())
        out.copy_(o)

    def max_over_ranks(self, values):
        import torch
        t = torch.tensor(list(values), dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.numpy()

    def barrier(self):
        self.dist.barrier()

    def close(self):
        self.dist.destroy_process_group()


def rccl_unique_id(rank, world):
    """The 128-byte communicator id of rank 0, handed over through the key-value store the launcher already runs
    (torch.distributed.run exports MASTER_ADDR / MASTER_PORT; TORCHELASTIC_USE_AGENT_STORE says the agent hosts it)."""
    from datetime import timedelta
    import torch.distributed as dist
    from pvsim import distributed as pd
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500"))
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    store = dist.TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=timedelta(seconds=600))
    key = "pvs_uid_" + os.environ.get("TORCHELASTIC_RUN_ID", "0") + "_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    if rank == 0:
        store.set(key, pd.new_unique_id())
    return bytes(store.get(key)), store


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # before anything initialises the HIP runtime (RCCL across processes)
    args = parse()
    if args.cpu_baseline_only:
        return cpu_baseline_child(args.images, args.total_queries or args.images, "fixed512" if args.workload == "corpus1m" else "ragged")
    if args.workload not in ("config2", "corpus1m"):
        return side_workload(args)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                      # no GPU call has happened in this process
    corpus1m = args.workload == "corpus1m"
    cpu = None
    if (not args.no_cpu_baseline and int(os.environ.get("RANK", "0")) == 0 and os.environ.get("PVS_BENCH_FORCE_DIST") != "1"
            and os.environ.get("PVS_BENCH_BACKEND", "rccl") != "gloo"):
        # rank 0 only, before anything touches the GPU (the child forks worker processes); the other ranks wait for it at the
        # communicator bootstrap
        q_cpu = args.total_queries if args.total_queries > 0 else (args.queries * int(os.environ.get("WORLD_SIZE", "1")) or args.images)
        cpu = cpu_baseline_subprocess(args.images, min(q_cpu, args.images) if not corpus1m else q_cpu, "fixed512" if corpus1m else "ragged")
    import torch
    import pvsim
    from pvsim import distributed as pd
    from pvsim.engine import DESC_F32, DESC_U8_ROOTSIFT

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (world == 1 and os.environ.get("PVS_BENCH_FORCE_DIST") == "1"):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: start `python3 bench.py --gpus N` without a launcher (it starts "
                         "its own ranks) or with torch.distributed.run --nproc-per-node N")
    # PVS_BENCH_BACKEND=gloo: REHEARSAL (ranks share the visible GPUs, host-staged collectives, see GlooStagedComm)
    backend = os.environ.get("PVS_BENCH_BACKEND", "rccl")
    # PVS_BENCH_FORCE_DIST=1 (started through torch.distributed.run with ONE rank): the multi-rank code path -- RCCL communicator
    # behind the C-ABI on its own stream, stream-ordered exchange, block-pair retrieval, list all-to-all -- on a single GPU: a
    # self-check of that path against the plain single-GPU retrieval, not a measurement
    forced = os.environ.get("PVS_BENCH_FORCE_DIST") == "1" and world == 1
    multi = world > 1 or forced
    if backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    overlap = os.environ.get("PVS_BENCH_OVERLAP", "1") != "0"      # exchange overlapped with the (r, r) block (RCCL only)

    # N > 1: the engine works on the stream torch allocates / fills on, so a step needs no host synchronisation between encode,
    # scoring and merge; the exchange has a context (= stream) of its own, ordered against the compute stream by events
    # (pvs_stream_wait), so the all-gather runs while the (r, r) block is scored
    comm, ctx_x, store = None, None, None
    if multi:
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
        ctx = pvsim.Context(local, stream=side.cuda_stream)
        if backend == "gloo":
            import torch.distributed as dist
            dist.init_process_group("gloo")
            comm = GlooStagedComm()
        else:
            ctx_x = pvsim.Context(local)
            uid, store = rccl_unique_id(rank, world)
            comm = pd.RcclComm(ctx_x, world, rank, uid)
    else:
        ctx = pvsim.Context(local)
    staged = backend == "gloo"
    tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"), allow_pickle=False)
    cb = ctx.codebook(tables["centroids"])
    if args.fused:
        from pvsim import _ffi
        ctx.set_option(_ffi.OPT_VLAD_PATH, _ffi.VLAD_PATH_FUSED)
    if args.prefilter_products != 3:
        from pvsim import _ffi
        ctx.set_option(_ffi.OPT_ASSIGN_PREFILTER, {2: 2, 1: 3}[args.prefilter_products])
    elif args.prefilter_16x16:
        from pvsim import _ffi
        ctx.set_option(_ffi.OPT_ASSIGN_PREFILTER, 4)

    # ---- corpus, sharded by image: rank r owns images [lo, hi)
    N = args.images
    per = (N + world - 1) // world
    lo, hi = min(N, rank * per), min(N, (rank + 1) * per)
    n_loc = hi - lo
    L = K_CLUSTERS * DIM
    k_top = 10 if corpus1m else TOPK
    retr = args.retrieval
    if retr == "f16" and not corpus1m:
        raise SystemExit("--retrieval f16 belongs to --workload corpus1m (BASELINE configs[4])")
    enc_loc = torch.empty((per, L), dtype=torch.float32, device=dev)       # padded to the common block size
    inv_loc = torch.ones((per,), dtype=torch.float32, device=dev)
    if n_loc < per:
        enc_loc[n_loc:].zero_()
    need_f32_all = multi and retr != "f16"
    enc_all = torch.empty((world * per, L), dtype=torch.float32, device=dev) if need_f32_all else enc_loc
    inv_all = torch.empty((world * per,), dtype=torch.float32, device=dev) if multi else inv_loc
    enc16_loc = enc16_all = None
    # Bounded query block on several GPUs (the N > 1 default): the QUERIES travel, the database stays.  Every rank all-gathers the
    # ranks' query blocks (world x nqB rows: 4.3 GB at 65536 queries) instead of the database blocks (65.5 GB at 10^6 images),
    # ranks all of them against its own database block, sends each owner its k candidates per query (all-to-all) and merges the
    # `world` lists it receives for its own queries.  Same flop per rank, a fifteenth of the bytes on the links.  All-vs-all
    # (queries = database) keeps the chunked database all-gather below.
    nqB = min(per, max(1, args.total_queries // world)) if args.total_queries > 0 else per
    travel = retr == "f16" and multi and args.total_queries > 0 and nqB * world < N
    if retr == "f16":
        enc16_loc = torch.empty((per, L), dtype=torch.float16, device=dev)
        if n_loc < per:
            enc16_loc[n_loc:].zero_()
        enc16_all = enc16_loc
        if travel:
            travel_bufs = [None]
            enc16_all = None
        elif multi:
            # the gathered fp16 corpus arrives in ROW CHUNKS of every rank's block (PVS_BENCH_XCHUNKS all-gathers instead of one),
            # so that the scoring of chunk c overlaps the transfer of chunk c + 1: [chunk][rank][rows of the chunk][L]
            x_chunks, x_pieces = pd.exchange_chunks(N, world, int(os.environ.get("PVS_BENCH_XCHUNKS", "4")))
            enc16_chunk = [torch.empty((world, c1 - c0, L), dtype=torch.float16, device=dev) for c0, c1 in x_chunks]
            enc16_all = None
    nq = n_loc if args.queries <= 0 else min(n_loc, args.queries)            # query rows of this rank (all of them by default)
    if args.total_queries > 0:                                               # strong scaling: the queries are shared out
        nq = min(n_loc, max(1, args.total_queries // world))
    idx = torch.empty((max(n_loc, nqB if args.total_queries > 0 else 0, 1), k_top), dtype=torch.int64, device=dev)
    val = torch.empty((idx.shape[0], k_top), dtype=torch.float32, device=dev)

    if corpus1m:
        # BASELINE configs[3]: every image has 512 raw SIFT-like uint8 descriptors, generated on the device CHUNK BY CHUNK
        # (the whole corpus of descriptors need not be resident: 65.5 GB at 1e6 images); a chunk is resident in HBM when
        # its encode is timed, the generation itself is not part of the step
        from pvsim import synth
        n_desc, CH = 512, 16384
        proto = torch.from_numpy(synth.sift_prototypes().astype(np.float32)).to(dev)
        raw = torch.empty((CH * n_desc, DIM), dtype=torch.uint8, device=dev)
        off = (torch.arange(CH + 1, device=dev, dtype=torch.int64) * n_desc).contiguous()
        kind, total_desc = DESC_U8_ROOTSIFT, n_loc * n_desc
        desc_bytes = total_desc * DIM

        def encode_all():
            """-> seconds spent encoding (generation excluded)"""
            t_enc = 0.0
            for c0 in range(0, n_loc, CH):
                cn = min(CH, n_loc - c0)
                g = torch.Generator(device=dev)
                g.manual_seed(1237 + 100003 * (lo + c0))             # a chunk's content depends on its global position only
                for s0 in range(0, cn * n_desc, 1 << 21):
                    e0 = min(cn * n_desc, s0 + (1 << 21))
                    z = torch.randint(0, proto.shape[0], (e0 - s0,), generator=g, device=dev)
                    x = proto[z] * torch.exp(0.35 * torch.randn((e0 - s0, DIM), generator=g, device=dev)) + \
                        4.8 * torch.rand((e0 - s0, DIM), generator=g, device=dev) ** 3
                    raw[s0:e0] = (x * (512.0 / x.norm(dim=1, keepdim=True).clamp_min(1e-9))).clamp_max(255.0).round().to(torch.uint8)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.vlad_encode_dev(cb, raw.data_ptr(), kind, off.data_ptr(), cn, cn * n_desc, enc_loc[c0:].data_ptr(),
                                    d_inv_norm=inv_loc[c0:].data_ptr())
                if retr == "f16":
                    ctx.f32_to_f16_dev(enc_loc[c0:].data_ptr(), cn * L, enc16_loc[c0:].data_ptr())
                ctx.sync()
                t_enc += time.perf_counter() - t0
                if rank == 0 and (c0 // CH) % 16 == 0:
                    print(f"[corpus1m] rank 0: {c0 + cn} / {n_loc} images encoded", file=sys.stderr, flush=True)
            return t_enc
    else:
        # BASELINE configs[1]: ragged images, fp32 RootSIFT rows (or raw uint8) resident in HBM; every rank generates only its
        # own shard (seeded per rank so shards differ)
        raw, offsets = make_corpus(n_loc, 1235 + 7919 * rank, dev)
        total_desc = int(offsets[-1])
        kind = DESC_U8_ROOTSIFT if args.desc == "u8" else DESC_F32
        desc = raw if args.desc == "u8" else rootsift_torch(raw)
        d_off = torch.from_numpy(offsets).to(dev)
        desc_bytes = total_desc * DIM * (1 if args.desc == "u8" else 4)

        def encode_all():
            ctx.vlad_encode_dev(cb, desc.data_ptr(), kind, d_off.data_ptr(), n_loc, total_desc, enc_loc.data_ptr(),
                                d_inv_norm=inv_loc.data_ptr())
            return None
    torch.cuda.synchronize()

    score_block = pd.device_score_block(ctx)
    ops = pd.DeviceOps(ctx, same_stream=multi)

    def to_exchange_stream():
        if ctx_x is not None:
            ctx_x.wait_for(ctx)

    def from_exchange_stream():
        if ctx_x is not None:
            ctx.wait_for(ctx_x)

    def a2a(out_t, in_t):
        to_exchange_stream()
        comm.all_to_all(out_t, in_t)
        from_exchange_stream()

    def new_tensor(shape, dtype, fill):
        return torch.full(shape, fill, dtype=getattr(torch, dtype), device=dev)

    def exchange_begin():
        """Start the exchange of the encoded blocks on the exchange stream (after the encode on the compute stream); returns the
        call that makes the compute stream wait for it.  The gloo rehearsal stages through the host and is synchronous."""
        to_exchange_stream()
        comm.all_gather(enc_loc, enc_all)
        comm.all_gather(inv_loc, inv_all)
        if staged or not overlap:
            from_exchange_stream()
            return lambda: None
        return from_exchange_stream

    filt_stats = [None]

    def retrieve(filtered_now):
        exchanged = exchange_begin() if (multi and retr != "f16") else (lambda: None)
        if travel:
            def score16(q, n_q, db, n_db, inv_q, inv_db, k_, off_, merge_, i_, v_):
                ctx.cosine_topk_f16_dev(q.data_ptr(), n_q, db.data_ptr(), n_db, L, inv_q.data_ptr(), inv_db.data_ptr(), k_, off_, merge_,
                                        i_.data_ptr(), v_.data_ptr())

            def travel_array(shape, dtype, fill):
                dt_ = getattr(torch, dtype) if isinstance(dtype, str) else dtype
                return torch.empty(shape, dtype=dt_, device=dev) if fill is None else torch.full(shape, fill, dtype=dt_, device=dev)

            travel_bufs[0] = pd.retrieve_traveling_queries(
                enc16_loc[:nqB], inv_loc[:nqB], enc16_loc, inv_loc, n_loc, lo, rank, world, k_top, comm, a2a, score16, ops.merge,
                travel_array, idx, val, gather_begin=to_exchange_stream, gather_arrived=from_exchange_stream, bufs=travel_bufs[0])
        elif retr == "f16" and multi:
            # The rank's own block first, from its local copy, while the first chunk arrives; then, chunk by chunk, the rows of
            # every other rank as soon as that chunk's all-gather has finished (the exchange stream runs ahead: the compute
            # stream waits only for what had been queued there when it asked).  Every rank takes part in every collective,
            # also one without rows or queries.
            to_exchange_stream()                             # the exchange stream waits for the encode, nothing later
            comm.all_gather(inv_loc, inv_all)
            comm.all_gather(enc16_loc[x_chunks[0][0]:x_chunks[0][1]], enc16_chunk[0])
            if not overlap or staged:
                for ci in range(1, len(x_chunks)):
                    comm.all_gather(enc16_loc[x_chunks[ci][0]:x_chunks[ci][1]], enc16_chunk[ci])
            if nq > 0:
                ctx.cosine_topk_f16_dev(enc16_loc.data_ptr(), nq, enc16_loc.data_ptr(), n_loc, L, inv_loc.data_ptr(), inv_loc.data_ptr(), k_top,
                                        lo, False, idx.data_ptr(), val.data_ptr())
            from_exchange_stream()                           # chunk 0 (or, without overlap, everything) has arrived
            for ci, (c0, c1) in enumerate(x_chunks):
                if overlap and not staged and ci + 1 < len(x_chunks):
                    comm.all_gather(enc16_loc[x_chunks[ci + 1][0]:x_chunks[ci + 1][1]], enc16_chunk[ci + 1])   # in flight under chunk ci's scoring
                for r in range(world):
                    g0, nv = x_pieces[ci][r]                 # global index of rank r's first row in this chunk, rows that exist
                    if r == rank or nv <= 0 or nq <= 0:
                        continue
                    ctx.cosine_topk_f16_dev(enc16_loc.data_ptr(), nq, enc16_chunk[ci][r].data_ptr(), nv, L, inv_loc.data_ptr(),
                                            inv_all.data_ptr() + g0 * 4, k_top, g0, True, idx.data_ptr(), val.data_ptr())
                if overlap and not staged and ci + 1 < len(x_chunks):
                    from_exchange_stream()
        elif retr == "f16":
            exchanged()
            ctx.cosine_topk_f16_dev(enc16_loc.data_ptr(), nq, enc16_all.data_ptr(), N, L, inv_loc.data_ptr(), inv_all.data_ptr(), k_top,
                                    0, False, idx.data_ptr(), val.data_ptr())
        elif filtered_now:
            exchanged()
            # this rank's queries against the gathered corpus through the fp16 prefilter + exact re-scoring: the same lists as the
            # exact path, no list exchange needed (the first N gathered rows are the real ones, padding only follows the last block)
            filt_stats[0] = ctx.cosine_topk_filtered_dev(enc_loc.data_ptr(), nq, enc_all.data_ptr(), N, L, inv_loc.data_ptr(),
                                                         inv_all.data_ptr(), k_top, idx.data_ptr(), val.data_ptr())
        elif not multi:
            ctx.cosine_topk_dev(enc_loc.data_ptr(), nq, enc_loc.data_ptr(), n_loc, L, inv_loc.data_ptr(), inv_loc.data_ptr(), k_top,
                                0, False, idx.data_ptr(), val.data_ptr())
        elif nq < n_loc:
            exchanged()       # a query subset: plain row-block x all-blocks scoring (the pair scheme needs every rank's full block)
            inv_m = inv_all.clone()
            inv_m[N:] = float("nan")
            ctx.cosine_topk_dev(enc_loc.data_ptr(), nq, enc_all.data_ptr(), world * per, L, inv_loc.data_ptr(), inv_m.data_ptr(), k_top,
                                0, False, idx.data_ptr(), val.data_ptr())
        else:
            # every block pair is scored once (dual-store GEMM), k-candidate lists exchanged, merged; the (r, r) block reads
            # the local copy, so it runs while the other blocks are still arriving
            i_, v_ = pd.retrieve_symmetric(enc_all, inv_all, N, rank, world, k_top, ops, a2a, new_tensor,
                                           own=(enc_loc, inv_loc), before_cross=exchanged)
            idx[:n_loc].copy_(i_)
            val[:n_loc].copy_(v_)

    def barrier():
        if ctx_x is not None:
            ctx_x.sync()
        ctx.sync()
        torch.cuda.synchronize()
        if multi:
            comm.barrier()

    def timed_steps(filtered_now):
        t_enc_sum = 0.0
        for _ in range(args.warmup):
            encode_all()
            retrieve(filtered_now)
        barrier()
        ctx.timers_enable(True)
        ctx.timers_reset()
        t0_ = time.perf_counter()
        t_gen_excl = 0.0
        for _ in range(args.steps):
            ta = time.perf_counter()
            te = encode_all()
            tb = time.perf_counter()
            if te is not None:                   # chunked corpus: only the encode part of the build counts
                t_gen_excl += (tb - ta) - te
                t_enc_sum += te
            retrieve(filtered_now)
        barrier()
        dt_ = time.perf_counter() - t0_ - t_gen_excl
        tm_ = ctx.timers()
        ctx.timers_enable(False)
        return dt_, tm_, t_enc_sum

    filtered = retr == "filtered"
    other = None
    if not multi and not corpus1m:
        # the variant that is NOT the headline of this run is timed first, its lists kept for the bit-for-bit comparison
        o_dt, o_tm, _ = timed_steps(not filtered)
        other = {"dt": o_dt, "timers": o_tm, "idx": idx.clone(), "val": val.clone(), "was_filtered": not filtered, "stats": filt_stats[0]}
    dt, timers, t_enc_total = timed_steps(filtered)
    if multi:
        dt = float(comm.max_over_ranks([dt])[0])
    ms_per_step = dt / args.steps * 1e3
    value = N / (dt / args.steps)

    # ---- sanity inside the bench: self-retrieval must return the image itself first
    got = idx[:nq, 0].cpu().numpy()
    assert np.array_equal(got, np.arange(lo, lo + nq)), "self-retrieval failed: top-1 is not the query image"
    check = None
    if (multi and (staged or forced)) or (corpus1m and retr != "exact"):
        # self-check: the multi-rank / filtered / fp16 result against the plain exact ranking of (a sample of) this rank's queries
        nchk = nq if retr != "f16" and (staged or forced) else min(nq, 256)
        ri = torch.empty((nchk, k_top), dtype=torch.int64, device=dev)
        rv = torch.empty((nchk, k_top), dtype=torch.float32, device=dev)
        barrier()
        exact_lists = False
        if multi and retr == "f16":
            ref_db, ref_n = None, 0
            if staged or forced:
                # The fp32 corpus was not gathered for the run and is not gathered for the check either (131 GB per rank at 10^6 images):
                # the exact fp32 lists of the sampled queries come from the same plan as the run's -- the sampled query blocks
                # travel, every rank ranks them EXACTLY against its own fp32 block, candidates return by all-to-all, the owner merges.
                nchkB = min(nqB if args.total_queries > 0 else per, 256)
                nchk = min(nq, nchkB)

                def score32(q, n_q, db, n_db, inv_q, inv_db, k_, off_, merge_, i_, v_):
                    ctx.cosine_topk_dev(q.data_ptr(), n_q, db.data_ptr(), n_db, L, inv_q.data_ptr(), inv_db.data_ptr(), k_, off_, merge_,
                                        i_.data_ptr(), v_.data_ptr())

                def chk_array(shape, dtype, fill):
                    dt_ = getattr(torch, dtype) if isinstance(dtype, str) else dtype
                    return torch.empty(shape, dtype=dt_, device=dev) if fill is None else torch.full(shape, fill, dtype=dt_, device=dev)

                ri = torch.empty((max(nchkB, 1), k_top), dtype=torch.int64, device=dev)
                rv = torch.empty((max(nchkB, 1), k_top), dtype=torch.float32, device=dev)
                pd.retrieve_traveling_queries(enc_loc[:nchkB], inv_loc[:nchkB], enc_loc, inv_loc, n_loc, lo, rank, world, k_top, comm, a2a,
                                              score32, ops.merge, chk_array, ri, rv, gather_begin=to_exchange_stream,
                                              gather_arrived=from_exchange_stream)
                barrier()
                ri, rv = ri[:nchk], rv[:nchk]
                exact_lists = True
        else:
            ref_db, ref_n = enc_all, (world * per if multi else n_loc)
        if ref_db is not None or exact_lists:
            if not exact_lists:
                inv_ref = inv_all.clone()
                if multi:
                    inv_ref[N:] = float("nan")                     # padding rows of the last block never rank
                torch.cuda.synchronize()
                ctx.cosine_topk_dev(enc_loc.data_ptr(), nchk, ref_db.data_ptr(), ref_n, L, inv_loc.data_ptr(), inv_ref.data_ptr(), k_top,
                                    0, False, ri.data_ptr(), rv.data_ptr())
            ctx.sync()
            if retr == "f16":
                a_, b_ = idx[:nchk].cpu().numpy(), ri.cpu().numpy()
                recall = float(np.mean([len(set(a_[i]) & set(b_[i])) / k_top for i in range(nchk)])) if nchk else 1.0
                assert recall >= 0.99, f"fp16 retrieval recall@{k_top} = {recall:.4f} against the exact ranking"
                check = {"fp16_recall_at_k_vs_exact_f32": round(recall, 5), "queries_checked": nchk}
            else:
                same = torch.equal(ri, idx[:nchk]) and torch.equal(rv.view(torch.int32), val[:nchk].view(torch.int32))
                assert same, f"rank {rank}: retrieval differs from the plain exact ranking"
                check = {"lists_bit_identical_to_plain_exact_ranking": True, "queries_checked": nchk}
                print(f"[rehearsal] rank {rank}: {nchk} queries identical to the single-GPU ranking", file=sys.stderr)
        else:
            print(f"[rehearsal] rank {rank}: fp16 lists, no fp32 corpus gathered (self-retrieval checked)", file=sys.stderr)

    if rank != 0:
        if comm is not None:
            comm.close()
        return

    # ---- roofline of the dominant kernel (the similarity GEMM): EXECUTED flop / measured launch duration.
    # Exact f32, single GPU or the pair scheme: the (r, r) block computes only the upper triangle of 128x128 tiles and mirrors
    # the rest, so the executed flop is ~half of the algorithmic 2*N*M*L of SURVEY.md section 8(d); `achieved` counts executed
    # flop only (never above peak).
    gemm_ms, gemm_n = timers["cosine_gemm"]
    f16_gemm = retr in ("f16", "filtered")
    t128 = (n_loc + 127) // 128
    launches_per_step = max(gemm_n / max(args.steps, 1), 1.0)
    if f16_gemm:
        alg_flop = 2.0 * nq * N * L
        flop_per_launch = alg_flop / launches_per_step          # panels of the fp16 GEMM (diagonal panels run symmetric: upper bound)
    elif not multi:
        alg_flop = 2.0 * nq * N * L
        flop_per_launch = 2.0 * 128 * 128 * L * (t128 * (t128 + 1) // 2) if nq == n_loc else alg_flop
    else:
        alg_flop = 2.0 * n_loc * N * L
        flop_per_launch = 2.0 * 128 * 128 * L * (t128 * (t128 + 1) / 2 + (world - 1) / 2.0 * t128 * t128) / launches_per_step
    gemm_avg_ms = gemm_ms / max(gemm_n, 1)
    achieved = flop_per_launch / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0
    peak = 2500.0 if f16_gemm else FP32_MFMA_PEAK_TFLOPS
    default_line = not multi and not corpus1m and N == 8189 and not filtered
    traffic, traffic_src = pmc_traffic("pvs::gemm_mfma_kernel", keep=lambda nm: not _gemm_is_f16(nm)) if default_line else (None, None)
    stages = {k: {"ms_total": round(v[0], 3), "launches": int(v[1]),
                  "ms_avg": round(v[0] / v[1], 4) if v[1] else None} for k, v in timers.items() if v[1]}
    enc_ms = (timers["assign"][0] + timers["aggregate"][0]) / args.steps
    enc_bytes = desc_bytes + n_loc * L * 4
    workload = (f"configs[3]{'/[4]' if retr == 'f16' else ''}: {N} images x 512 uint8 SIFT-like descriptors (fused RootSIFT), VLAD K=256 encode"
                f" (chunks of 16384 images, generation untimed) + {nq * world if multi else nq} x {N} cosine + top-{k_top}" if corpus1m else
                f"configs[1]: {N} images x ragged SIFT-like descriptors (mean {total_desc / max(n_loc, 1):.0f}/image),"
                f" D=128, VLAD K=256 encode + {N}x{N} cosine + top-{k_top}")
    out = {
        "metric": "images/sec encoded + top-k retrieved, VLAD K256 RootSIFT",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        **({"exchange": "allgather over " + ("gloo (host-staged REHEARSAL)" if staged else "RCCL behind the C-ABI (" + pd.RcclComm.library() + ")"),
            "exchange_overlaps_own_block": bool(overlap and not staged and retr in ("exact", "f16")),
            "exchange_plan": (f"queries travel: {world} x {nqB} fp16 query rows all-gathered ({world * nqB * L * 2 / 1e9:.2f} GB), every rank ranks them "
                              f"against its own database block, k candidates per query back by all-to-all, merged by the owner" if travel else
                              "database travels: every rank all-gathers all encoding blocks and ranks its own queries against them")} if multi else {}),
        **({"backend": "ONE-rank RCCL self-check of the multi-rank path: not a measurement"} if forced else {}),
        **({"backend": "gloo REHEARSAL (ranks share GPUs, host-staged collectives): not a measurement"} if staged else {}),
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "queries_per_step": int(nq * world if multi else nq),
        "dtype": "f16" if retr == "f16" else "f32", "data": "synthetic", "retrieval": retr,
        "config": {"workload": workload, "images": N, "descriptors_rank0": total_desc, "descriptor_rows": "u8" if corpus1m else args.desc,
                   "K": K_CLUSTERS, "D": DIM, "topk": k_top, "parallelism": f"image-sharded x{world}",
                   "encode_path": "fused one-read kernel" if args.fused else "assign + aggregate"},
        "roofline": {"kernel": ("gemm_f16_8ph_kernel<256x256, 16x16x32> (fp16 operands, fp32 accumulate, 8-phase schedule)" if retr == "f16" else
                                "gemm_f16_2lvl_kernel<256x128, chains of 1024 k> / gemm_mfma_kernel<128,128,f16,two-level> (prefilter GEMM of the filtered retrieval)" if filtered else
                                "gemm_mfma_kernel<128,128,f32> (cosine GEMM: main + split-K tail)"), "bound": "mfma", "achieved": round(achieved, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "flop_per_launch": flop_per_launch,
                     "mfma_pipe_busy_frac": pmc_mfma_busy("pvs::gemm_mfma_kernel", keep=lambda nm: not _gemm_is_f16(nm)) if default_line else None,
                     "algorithmic_flop_per_launch": alg_flop,
                     "algorithmic_equiv_TFLOPs": round(alg_flop / (gemm_avg_ms * launches_per_step * 1e-3) / 1e12, 2) if gemm_n else None,
                     "avg_launch_ms": round(gemm_avg_ms, 4)},
        "stages": stages,
        "encode": {"ms_per_step": round(enc_ms, 3), "images_per_s": round(n_loc / (enc_ms * 1e-3), 1) if enc_ms else None,
                   "algorithmic_GBps": round(enc_bytes / (enc_ms * 1e-3) / 1e9, 1) if enc_ms else None,
                   "hbm_roofline_frac": round(enc_bytes / (enc_ms * 1e-3) / 1e9 / 8000.0, 4) if enc_ms else None,
                   "assign_TFLOPs": round(2.0 * total_desc * K_CLUSTERS * DIM / (timers["assign"][0] / args.steps * 1e-3) / 1e12, 2)
                   if timers["assign"][0] else None},
        "device": ctx.device_name(),
    }
    if corpus1m:
        out["corpus_build"] = {"encode_s_per_step": round(t_enc_total / args.steps, 3), "resident_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    if check is not None:
        out["self_check"] = check
    if filt_stats[0] and (filtered or retr == "filtered"):
        out["filter_stats"] = filt_stats[0]

    if other is not None:
        same_lists = bool(torch.equal(other["idx"], idx)) and bool(torch.equal(other["val"].view(torch.int32), val.view(torch.int32)))
        f_dt, f_tm, f_st = (other["dt"], other["timers"], other["stats"]) if other["was_filtered"] else (dt, timers, filt_stats[0])
        e_dt = dt if other["was_filtered"] else other["dt"]
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

    if args.pcie and not multi and not corpus1m:
        h_desc = desc.cpu().numpy()
        t1 = time.perf_counter()
        v = ctx.vlad_encode(cb, h_desc, offsets, kind)
        ctx.cosine_topk(v, v, TOPK)
        out["pcie_inclusive_images_per_s"] = round(N / (time.perf_counter() - t1), 1)

    if cpu is not None:
        out["cpu_baseline"] = cpu       # single-process baseline of the same workload, timed on rank 0's host cores
    if multi and not (staged or forced):
        base = strong_scaling_base(args, retr)
        if base is not None:
            out["strong_scaling_base_1gpu"] = base

    print(json.dumps(out))
    if comm is not None:
        comm.close()


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
    counts = synth.ragged_counts(n_img, seed) if _W.get("gen", "ragged") == "ragged" else [512] * n_img
    rng = np.random.default_rng(seed)
    raws = [synth.sift_like(int(c), rng) for c in counts]
    t0 = time.perf_counter()
    v = np.vstack([orc.vlad_encode_one(orc.rootsift(r), _W["C"]) for r in raws])
    return v, time.perf_counter() - t0, int(sum(counts))


def _w_query(job):
    lo, n = job
    orc, db = _W["orc"], _W["db"]
    t0 = time.perf_counter()
    for i in range(lo, lo + n):
        orc.retrieve_top_k(db[i % len(db)], db, TOPK)
    return time.perf_counter() - t0


def _host_cpus():
    """-> (cpus this process may run on, the cgroup's CPU quota in cores or None)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except (OSError, ValueError):
        pass
    return avail, quota


def cpu_baseline_child(n_full, n_queries, gen):
    """Runs in the child process: prints one JSON object.  n_full images in the corpus, n_queries of them queried per step;
    gen = "ragged" (configs[1]) or "fixed512" (configs[3]: 512 descriptors per image)."""
    import multiprocessing as mp
    import platform
    tables = np.load(os.path.join(REPO, "tests", "golden", "tables_k256_d128.npz"), allow_pickle=False)
    C = np.ascontiguousarray(tables["centroids"], dtype=np.float32)
    cores, quota = _host_cpus()
    # BASELINE.md section 3, mode B: os.cpu_count() workers -- the CPUs this process may actually use (affinity mask, and the
    # cgroup quota where one is set: more workers than that only time-slice)
    P = max(1, cores if quota is None else min(cores, int(np.ceil(quota))))
    _W["gen"] = gen
    # ---- mode B first (it also builds the database): P single-thread workers
    ctx = mp.get_context("fork")
    per = max(2, -(-CPU_DB_ROWS // P))
    jobs = [(9000 + i, per) for i in range(P)]
    t0 = time.perf_counter()
    with ctx.Pool(P, initializer=_w_init, initargs=(C,)) as pool:
        parts = pool.map(_w_encode, jobs)
    wall_b_enc = time.perf_counter() - t0
    db = np.ascontiguousarray(np.vstack([p[0] for p in parts])[:max(CPU_DB_ROWS, 8 * P)])
    n_b = db.shape[0]
    mean_desc = sum(p[2] for p in parts) / (per * P)
    # wall time includes the worker start-up; the workers' own clocks give the steady-state rate
    enc_b = max(p[1] for p in parts) / per / P * 1.0      # seconds per image at P-way throughput
    _W["db"] = db
    q_per = 8
    qjobs = [((i * q_per) % n_b, q_per) for i in range(P)]
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
    qf = n_queries / n_full                               # queries per image of the corpus in one step
    phys = None
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        ids = set()
        cur = {}
        for line in txt.splitlines():
            if ":" in line:
                k_, v_ = [x.strip() for x in line.split(":", 1)]
                cur[k_] = v_
            elif cur:
                ids.add((cur.get("physical id"), cur.get("core id")))
                cur = {}
        phys = len(ids) if ids else None
        names = [l.split(":", 1)[1].strip() for l in txt.splitlines() if l.startswith("model name")]
    except OSError:
        names = []
    out = {"value": round(1.0 / (enc_b + qf * ret_b), 2), "unit": "images/s", "cores": P, "kind": "port",
           "sample": f"NumPy restatement of the reference's procedure (oracle/pvsim_oracle.py). Mode B (value): {P} worker processes, 1 BLAS "
                     f"thread each: {per * P} images encoded (mean {mean_desc:.0f} descriptors), {P * q_per} queries ranked against a "
                     f"{n_b}-row database; per-query cost scaled x{n_full / n_b:.2f} to the {n_full}-row database, {n_queries} queries per "
                     f"{n_full}-image step. Mode A: one process, library-default threads: {n_a} images encoded, {nq_a} queries",
           "mode_B": {"images_per_s": round(1.0 / (enc_b + qf * ret_b), 2), "encode_images_per_s": round(1.0 / enc_b, 1),
                      "retrieve_queries_per_s": round(1.0 / ret_b, 3), "workers": P, "threads_per_worker": 1, "queries_per_worker": q_per,
                      "encode_wall_s_incl_startup": round(wall_b_enc, 2), "query_wall_s": round(wall_b_q, 2)},
           "mode_A": {"images_per_s": round(1.0 / (enc_a + qf * ret_a), 2), "encode_images_per_s": round(1.0 / enc_a, 1),
                      "retrieve_queries_per_s": round(1.0 / ret_a, 3), "threadpools": pools},
           "host": {"cpus_available": cores, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(), "physical_cores": phys,
                    "machine": platform.processor() or platform.machine(), "numpy": np.__version__,
                    "cpu_model": names[0] if names else None}}
    print(json.dumps(out))


def cpu_baseline_subprocess(n_full, n_queries=None, gen="ragged"):
    """Start the CPU baseline as a child process.  Must be called BEFORE this process initialises the GPU."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--images", str(n_full),
                        "--total-queries", str(n_full if n_queries is None else n_queries), "--workload",
                        "corpus1m" if gen == "fixed512" else "config2"],
                       capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        return {"error": r.stderr[-800:]}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": "no output"}


def strong_scaling_base(args, retr):
    """The committed one-GPU run of the SAME workload and flags (profiles/rNN_corpus1m_base_bench.json), so that an N > 1 line
    carries the base its strong scaling is measured from (the driver's own N = 1 run is the configs[1] headline)."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_corpus1m_base_bench.json")), reverse=True):
        try:
            d = json.load(open(f))
            if (d["config"]["images"] == args.images and d.get("retrieval") == retr and d.get("n_gpus") == 1
                    and d.get("queries_per_step") == max(1, args.total_queries)):
                return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "source": os.path.basename(f)}
        except Exception:
            continue
    return None


if __name__ == "__main__":
    main()
