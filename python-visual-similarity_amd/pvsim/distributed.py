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

__all__ = ["shard_range", "gather_blocks", "mask_padding", "retrieve_sharded", "device_score_block", "ShardedVLADIndex"]


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
