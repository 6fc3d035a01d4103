"""Multi-GPU path: one process per GPU, images sharded contiguously, ONE exchange (all-gather of the encoding
blocks over RCCL/xGMI via torch.distributed), then every rank ranks its own query block against all blocks.

    rank r owns images [r*B, min(N, (r+1)*B)),  B = ceil(N / world)       (global index = r*B + local index,
                                                                           i.e. dict insertion order, eval.py:28)

torch is plumbing here (device tensors + the collective).  The scoring itself is ONE C-ABI call
`pvs_cosine_topk_dev` per rank (local queries x whole gathered corpus: a GEMM that fills the chip instead of
`world` small ones).  Blocks have equal size B, so a row's position in the gathered array is its true global index;
the trailing padding rows of the last block(s) carry a NaN inverse norm and therefore NaN scores, which the select
kernel ranks last.  The logic is written against a `score_block` callable so that the same host code runs under
`gloo` on CPU in tests/test_dist_gloo.py.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

__all__ = ["shard_range", "gather_blocks", "mask_padding", "retrieve_sharded", "device_score_block",
           "retrieve_symmetric", "symmetric_local", "symmetric_finish", "DeviceOps", "ShardedVLADIndex"]


def shard_range(n_total: int, world: int, rank: int) -> tuple[int, int, int]:
    """-> (lo, hi, block) with block = ceil(n_total / world); ranks past the end own nothing."""
    block = (n_total + world - 1) // world
    lo = min(n_total, rank * block)
    return lo, min(n_total, lo + block), block


def gather_blocks(enc_loc, inv_loc, group=None):
    """all_gather_into_tensor of the (block, L) encodings and (block,) inverse norms -> (world*block, L), (world*block,).
    `enc_loc` must already be padded to the common block size (padding rows are ignored by retrieve_sharded)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    enc_all = torch.empty((world * enc_loc.shape[0], enc_loc.shape[1]), dtype=enc_loc.dtype, device=enc_loc.device)
    inv_all = torch.empty((world * inv_loc.shape[0],), dtype=inv_loc.dtype, device=inv_loc.device)
    dist.all_gather_into_tensor(enc_all, enc_loc.contiguous(), group=group)
    dist.all_gather_into_tensor(inv_all, inv_loc.contiguous(), group=group)
    return enc_all, inv_all


def mask_padding(inv_loc, n_loc: int) -> None:
    """Mark the padding rows of a rank's block (rows >= n_loc) with a NaN inverse norm IN PLACE, before the
    all-gather: their scores become NaN, which the select kernel ranks below every number, so they can never
    displace a real image.  Because every block has the same size B and only trailing rows are padding, the row
    position in the gathered (world*B, L) array IS the true global image index."""
    if n_loc < inv_loc.shape[0]:
        inv_loc[n_loc:] = float("nan")


def retrieve_sharded(enc_loc, inv_loc, enc_all, inv_all, n_total: int, rank: int, world: int, k: int,
                     score_block: Callable, idx, val):
    """Rank-local queries (rows [0, n_loc) of enc_loc) against the whole gathered corpus in ONE scoring call.

    score_block(q, n_q, db, n_db, inv_q, inv_db, k, col_offset, merge, idx, val).  `inv_all` must carry NaN on
    padding rows (mask_padding before the gather).  Returns the number of local queries."""
    lo, hi, block = shard_range(n_total, world, rank)
    n_loc = hi - lo
    if n_loc == 0:
        return 0
    if world == 1:
        score_block(enc_loc, n_loc, enc_all, n_loc, inv_loc, inv_all, k, 0, False, idx, val)   # symmetric path
    else:
        score_block(enc_loc, n_loc, enc_all, world * block, inv_loc, inv_all, k, 0, False, idx, val)
    return n_loc


def device_score_block(ctx):
    """score_block for CUDA tensors: one pvs_cosine_topk_dev call (GEMM panel + select, running-list merge)."""

    def score(q, n_q, db, n_db, inv_q, inv_db, k, col_offset, merge, idx, val):
        ctx.cosine_topk_dev(q.data_ptr(), n_q, db.data_ptr(), n_db, q.shape[1], inv_q.data_ptr(), inv_db.data_ptr(),
                            k, col_offset, merge, idx.data_ptr(), val.data_ptr())

    return score


# ------------------------------------------------------------------------------------------------------------------
# Symmetric all-vs-all retrieval: cos(a, b) = cos(b, a), so every unordered block pair is scored ONCE, by one of its
# two ranks, with a GEMM that stores the panel AND its transpose (pvs_cosine_dual_dev).  The row-wise top-k of the
# panel serves the computing rank's queries, the row-wise top-k of the transposed panel serves the partner's
# queries and is shipped to it in one all-to-all of k-candidate lists (B*k*12 bytes per pair); every rank finally
# merges the lists it computed with the lists it received (pvs_topk_merge_dev).
#
#   rank r scores   (r, r)                  symmetric kernel (upper-triangle tiles, mirrored)
#                   (r, r+1 .. r+h)         h = (P-1)//2 full blocks, dual store, in ONE launch (two on wrap-around)
#                   (r, r+P/2)  if P even   half of that block: the lower rank takes the partner's rows [0, B/2),
#                                           the higher rank the lower rank's rows [B/2, B) -- see _half_split
#   => N^2 / (2P) scores per rank instead of N^2 / P.
# Results are bit-identical to the single-GPU path: a score does not depend on which tile, launch or K-split produced
# it (csrc/gemm_mfma.hpp), the transposed store copies the same fp32 value, and all lists are ordered
# (score desc, index asc).
class DeviceOps:
    """The four device operations of the scheme, bound to a pvsim.Context (tests plug in CPU stand-ins)."""

    def __init__(self, ctx, same_stream: bool = False):
        """same_stream=True: the context was created on the stream the caller's tensors / collectives use (e.g.
        pvsim.Context(dev, stream=torch.cuda.current_stream().cuda_stream)), so no host synchronisation is needed
        between this object's launches and the caller's."""
        self.ctx = ctx
        self.same_stream = same_stream

    def sym_topk(self, q, n, inv, k, col_offset, idx, val):
        self.ctx.cosine_topk_dev(q.data_ptr(), n, q.data_ptr(), n, q.shape[1], inv.data_ptr(), inv.data_ptr(), k,
                                 col_offset, False, idx.data_ptr(), val.data_ptr())

    def dual(self, a, m, b, n, inv_a, inv_b, panel, panel_t):
        self.ctx.cosine_dual_dev(a.data_ptr(), m, b.data_ptr(), n, a.shape[1], inv_a.data_ptr(), inv_b.data_ptr(),
                                 panel.data_ptr(), n, panel_t.data_ptr(), m)

    def topk(self, scores, nq, ncols, k, col_offset, merge, idx, val):
        self.ctx.topk_dev(scores.data_ptr(), nq, ncols, ncols, k, col_offset, merge, idx.data_ptr(), val.data_ptr())

    def merge(self, idx_lists, val_lists, n_lists, nq, k, idx, val):
        self.ctx.topk_merge_dev(idx_lists.data_ptr(), val_lists.data_ptr(), n_lists, nq, k, idx.data_ptr(), val.data_ptr())

    def sync(self):
        if not self.same_stream:
            self.ctx.sync()


def _rows(n_total, world, r):
    lo, hi, _ = shard_range(n_total, world, r)
    return hi - lo


def retrieve_symmetric(enc_all, inv_all, n_total: int, rank: int, world: int, k: int, ops, all_to_all, new_tensor,
                       own=None, before_cross=None):
    """Top-k of this rank's queries against the whole corpus, scoring each block pair once (see above).

    own = (enc_loc, inv_loc) and before_cross (a callable) let the exchange overlap the (r, r) block: that block is scored from
    the rank's local copy, then before_cross() must make the gathered rows of the OTHER ranks visible (e.g. wait on the
    asynchronous all-gather) before the cross blocks are scored.

    enc_all (world*B, L) / inv_all (world*B,): the gathered encodings and inverse norms (padding rows unused).
    ops: DeviceOps-like.  all_to_all(out, inp): exchange of equal (B*k)-element slabs between ranks, e.g.
    torch.distributed.all_to_all_single.  new_tensor(shape, dtype, fill): allocator on the right device.
    Returns (idx (n_loc, k) int64, val (n_loc, k) float32)."""
    st = symmetric_local(enc_all, inv_all, n_total, rank, world, k, ops, new_tensor, own=own, before_cross=before_cross)
    all_to_all(st["m_idx"][:world].reshape(-1), st["s_idx"].reshape(-1))
    all_to_all(st["m_val"][:world].reshape(-1), st["s_val"].reshape(-1))
    return symmetric_finish(st, ops, new_tensor)


def symmetric_local(enc_all, inv_all, n_total: int, rank: int, world: int, k: int, ops, new_tensor, own=None,
                    before_cross=None) -> dict:
    """Phase 1 (no communication of its own): score this rank's block pairs; returns the send / merge buffers.
    own / before_cross: see retrieve_symmetric."""
    import torch
    lo, hi, B = shard_range(n_total, world, rank)
    n_r = hi - lo
    P = world
    # slot p < P: list computed by rank p for my queries (received); slot P: what I computed for myself
    m_idx = new_tensor((P + 1, B, k), torch.int64, -1)
    m_val = new_tensor((P + 1, B, k), torch.float32, float("-inf"))
    s_idx = new_tensor((P, B, k), torch.int64, -1)           # slot p: list I computed for rank p's queries
    s_val = new_tensor((P, B, k), torch.float32, float("-inf"))
    own_i, own_v = m_idx[P], m_val[P]
    blk = lambda t, r: t[r * B:(r + 1) * B]                  # noqa: E731

    if n_r > 0:
        # ---- (r, r): symmetric kernel
        if own is not None:
            ops.sym_topk(own[0], n_r, own[1], k, rank * B, own_i, own_v)
        else:
            ops.sym_topk(blk(enc_all, rank), n_r, blk(inv_all, rank), k, rank * B, own_i, own_v)
    if before_cross is not None:                             # every rank, also one without rows: the exchange is collective
        before_cross()
    if n_r > 0:
        # ---- (r, r+1 .. r+h): full blocks, dual store; contiguous runs of blocks go in one launch
        h = (P - 1) // 2
        # contiguous runs of partner blocks (one launch each): a run ends at the wrap-around, after a short block
        # (only the last non-empty block of the corpus can be short) and skips empty blocks
        runs, cur = [], None
        for j in range(1, h + 1):
            s_ = (rank + j) % P
            n_s = _rows(n_total, P, s_)
            if n_s == 0 or (cur is not None and s_ != cur[0] + cur[1]):
                if cur is not None:
                    runs.append(tuple(cur))
                    cur = None
                if n_s == 0:
                    continue
            if cur is None:
                cur = [s_, 0]
            cur[1] += 1
            if n_s < B:
                runs.append(tuple(cur))
                cur = None
        if cur is not None:
            runs.append(tuple(cur))
        for s0, length in runs:
            n_cols = sum(_rows(n_total, P, s0 + t) for t in range(length))
            if n_cols == 0:
                continue
            # blocks s0 .. s0+length-1 are contiguous in enc_all; only the LAST rank's block can be short, and a
            # short block inside a run would leave a gap -- runs therefore never extend past a short block
            span = (length - 1) * B + _rows(n_total, P, s0 + length - 1)
            assert span == n_cols, "a short block may only end a run"
            panel = new_tensor((n_r, span), torch.float32, 0.0)
            panel_t = new_tensor((span, n_r), torch.float32, 0.0)
            ops.dual(blk(enc_all, rank), n_r, enc_all[s0 * B:], span, blk(inv_all, rank), inv_all[s0 * B:], panel, panel_t)
            ops.topk(panel, n_r, span, k, s0 * B, True, own_i, own_v)                       # my queries
            for t in range(length):                                                         # partners' queries
                s, n_s = s0 + t, _rows(n_total, P, s0 + t)
                if n_s > 0:
                    ops.topk(panel_t[t * B:], n_s, n_r, k, rank * B, False, s_idx[s], s_val[s])
        # ---- (r, r + P/2) for even P: the block pair is split between its two ranks
        if P % 2 == 0:
            partner = (rank + P // 2) % P
            a, b = min(rank, partner), max(rank, partner)    # panel = Q_a x DB_b^T; a takes b's rows [0, hb), b the rest
            n_a, n_b = _rows(n_total, P, a), _rows(n_total, P, b)
            hb = min(n_b, (B + 1) // 2)
            c0, c1 = (0, hb) if rank == a else (hb, n_b)     # columns of the panel (= rows of block b) I compute
            if n_a > 0 and c1 > c0:
                w = c1 - c0
                panel = new_tensor((n_a, w), torch.float32, 0.0)
                panel_t = new_tensor((w, n_a), torch.float32, 0.0)
                ops.dual(blk(enc_all, a), n_a, enc_all[b * B + c0:], w, blk(inv_all, a), inv_all[b * B + c0:], panel, panel_t)
                if rank == a:
                    ops.topk(panel, n_a, w, k, b * B + c0, True, own_i, own_v)              # my queries vs b[0:hb)
                    ops.topk(panel_t, w, n_a, k, a * B, False, s_idx[b][c0:], s_val[b][c0:])   # b's queries [0:hb) vs me
                else:
                    ops.topk(panel, n_a, w, k, b * B + c0, False, s_idx[a], s_val[a])       # a's queries vs my rows [hb:)
                    ops.topk(panel_t, w, n_a, k, a * B, True, own_i[c0:], own_v[c0:])       # my queries [hb:) vs a
    ops.sync()
    return {"m_idx": m_idx, "m_val": m_val, "s_idx": s_idx, "s_val": s_val, "n_r": n_r, "B": B, "k": k, "P": P}


def symmetric_finish(st: dict, ops, new_tensor):
    """Phase 2 (after the all-to-all filled m_idx[:P] / m_val[:P]): merge the P received lists with the own list."""
    import torch
    out_i = new_tensor((st["B"], st["k"]), torch.int64, -1)
    out_v = new_tensor((st["B"], st["k"]), torch.float32, float("-inf"))
    if st["n_r"] > 0:
        ops.merge(st["m_idx"], st["m_val"], st["P"] + 1, st["B"], st["k"], out_i, out_v)
        ops.sync()
    return out_i[:st["n_r"]], out_v[:st["n_r"]]


class ShardedVLADIndex:
    """Encode this rank's images on its GPU, exchange once, answer all-vs-all top-k for the local block."""

    def __init__(self, ctx, codebook, n_total: int, group=None):
        import torch.distributed as dist
        self.ctx, self.cb, self.n_total, self.group = ctx, codebook, n_total, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi, self.block = shard_range(n_total, self.world, self.rank)

    def encode_local(self, d_desc, kind, d_offsets, total_desc, power=1.0, norm_order=2, epsilon=1e-9):
        """d_desc / d_offsets: CUDA tensors holding this rank's packed descriptors and CSR offsets."""
        import torch
        n_loc = self.hi - self.lo
        L = self.cb.K * self.cb.D
        dev = d_desc.device
        self.enc_loc = torch.zeros((self.block, L), dtype=torch.float32, device=dev)
        self.inv_loc = torch.ones((self.block,), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        self.ctx.vlad_encode_dev(self.cb, d_desc.data_ptr(), kind, d_offsets.data_ptr(), n_loc, total_desc,
                                 self.enc_loc.data_ptr(), power, norm_order, epsilon,
                                 d_inv_norm=self.inv_loc.data_ptr())
        self.ctx.sync()
        if self.world > 1:
            mask_padding(self.inv_loc, n_loc)
        return self.enc_loc

    def exchange(self):
        if self.world > 1:
            self.enc_all, self.inv_all = gather_blocks(self.enc_loc, self.inv_loc, self.group)
            import torch
            torch.cuda.synchronize(self.enc_loc.device)
        else:
            self.enc_all, self.inv_all = self.enc_loc, self.inv_loc

    def topk(self, k: int):
        import torch
        n_loc = self.hi - self.lo
        dev = self.enc_loc.device
        idx = torch.full((max(n_loc, 1), k), -1, dtype=torch.int64, device=dev)
        val = torch.full((max(n_loc, 1), k), float("-inf"), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        retrieve_sharded(self.enc_loc, self.inv_loc, self.enc_all, self.inv_all, self.n_total, self.rank, self.world,
                         k, device_score_block(self.ctx), idx, val)
        self.ctx.sync()
        return idx[:n_loc], val[:n_loc]
