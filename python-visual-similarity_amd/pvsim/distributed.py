"""Multi-GPU path: one process per GPU, images sharded contiguously, ONE exchange of the encoding blocks (RCCL over xGMI,
behind the C-ABI: pvs_comm_* in include/pvsim.h), then every rank ranks its own query block against all blocks.

    rank r owns images [r*B, min(N, (r+1)*B)),  B = ceil(N / world)       (global index = r*B + local index,
                                                                           i.e. dict insertion order, eval.py:28)

Nothing here needs torch: device memory is pvs_malloc memory wrapped in `DevArray`, the collectives are `RcclComm`.  The
retrieval logic only slices / indexes its array arguments along the first axis and asks them for `.data_ptr()`, so it runs
unchanged on DevArrays, on torch CUDA tensors (bench.py generates its corpus with torch) and -- with CPU stand-ins for the
device operations -- on torch CPU tensors under `gloo` (tests/test_dist_gloo.py).

Blocks have equal size B, so a row's position in the gathered array is its true global index; the trailing padding rows of
the last block(s) carry a NaN inverse norm and therefore NaN scores, which the select kernel ranks last.
"""
from __future__ import annotations

import ctypes as C
import socket
import struct
import time
from typing import Callable

import numpy as np

__all__ = ["shard_range", "gather_blocks", "mask_padding", "retrieve_sharded", "device_score_block",
           "retrieve_symmetric", "symmetric_local", "symmetric_finish", "DeviceOps", "ShardedVLADIndex",
           "DevArray", "DevicePool", "RcclComm", "new_unique_id", "exchange_unique_id"]

_ITEM = {"float32": 4, "int64": 8, "int32": 4, "float64": 8, "float16": 2, "uint8": 1}


def shard_range(n_total: int, world: int, rank: int) -> tuple[int, int, int]:
    """-> (lo, hi, block) with block = ceil(n_total / world); ranks past the end own nothing."""
    block = (n_total + world - 1) // world
    lo = min(n_total, rank * block)
    return lo, min(n_total, lo + block), block


def exchange_chunks(n_total: int, world: int, n_chunks: int):
    """Row chunks of the block exchange (the fp16 retrieval scores chunk c while chunk c + 1 travels): every rank's block of
    B = ceil(n_total / world) rows is cut at the same local offsets.  -> (chunks [(c0, c1) local row ranges], pieces) where
    pieces[ci][r] = (global index of the first row, number of rows that exist) of rank r's part of chunk ci (0 rows past
    n_total: only the last non-empty block is short).  All parts together tile [0, n_total) exactly once."""
    _, _, block = shard_range(n_total, world, 0)
    n_chunks = max(1, min(int(n_chunks), max(block, 1)))
    per_c = -(-block // n_chunks)
    chunks = [(c * per_c, min(block, (c + 1) * per_c)) for c in range(n_chunks) if c * per_c < block]
    pieces = [[(r * block + c0, max(0, min(n_total, r * block + c1) - (r * block + c0))) for r in range(world)] for c0, c1 in chunks]
    return chunks, pieces


# ------------------------------------------------------------------------------------------------------------------
# device memory without torch
class DevArray:
    """A C-contiguous n-d view of device memory: first-axis indexing / slicing, reshape(-1), data_ptr() -- the subset of
    the tensor protocol the retrieval logic uses."""

    def __init__(self, ctx, ptr: int, shape, dtype: str, owner=None):
        self.ctx, self.ptr, self.shape, self.dtype, self._owner = ctx, int(ptr), tuple(int(s) for s in shape), dtype, owner

    @property
    def itemsize(self) -> int:
        return _ITEM[self.dtype]

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.itemsize

    def data_ptr(self) -> int:
        return self.ptr

    def _row_bytes(self) -> int:
        return int(np.prod(self.shape[1:], dtype=np.int64)) * self.itemsize

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, step = key.indices(self.shape[0])
            if step != 1:
                raise IndexError("DevArray: unit stride only")
            stop = max(stop, start)
            return DevArray(self.ctx, self.ptr + start * self._row_bytes(), (stop - start,) + self.shape[1:], self.dtype, self)
        i = int(key)
        if i < 0:
            i += self.shape[0]
        if not 0 <= i < self.shape[0]:
            raise IndexError("DevArray index out of range")
        return DevArray(self.ctx, self.ptr + i * self._row_bytes(), self.shape[1:], self.dtype, self)

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        n = int(np.prod(self.shape, dtype=np.int64))
        if shape.count(-1) > 1:
            raise ValueError("DevArray.reshape: one -1 at most")
        if -1 in shape:
            known = int(np.prod([d for d in shape if d != -1], dtype=np.int64)) if len(shape) > 1 else 1
            shape = tuple(n // max(known, 1) if d == -1 else int(d) for d in shape)
        if int(np.prod(shape, dtype=np.int64)) != n:
            raise ValueError("DevArray.reshape: size mismatch")
        return DevArray(self.ctx, self.ptr, shape, self.dtype, self)

    def fill(self, value):
        n = int(np.prod(self.shape, dtype=np.int64))
        if n:
            self.ctx.fill_dev(self.ptr, n, self.dtype, value)
        return self

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError("DevArray.upload: size mismatch")
        from . import _ffi
        _ffi.check(_ffi.lib().pvs_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), _ffi.ptr(a), a.nbytes))
        return self

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            from . import _ffi
            _ffi.check(_ffi.lib().pvs_memcpy_d2h(self.ctx.handle, _ffi.ptr(out), C.c_void_p(self.ptr), out.nbytes))
        return out


class DevicePool:
    """Grow-only cache of pvs_malloc blocks keyed by size class.  Every user enqueues on the context's one stream, so a block
    handed out again after release_all() is reused in stream order (hipMalloc / hipFree synchronise the device: never per step)."""

    def __init__(self, ctx):
        self.ctx, self._free, self._used = ctx, {}, []

    def empty(self, shape, dtype: str) -> DevArray:
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * _ITEM[dtype], 16)
        cls = 1 << (nbytes - 1).bit_length()
        lst = self._free.setdefault(cls, [])
        buf = lst.pop() if lst else self.ctx.buffer(cls)
        self._used.append((cls, buf))
        return DevArray(self.ctx, buf.ptr, shape, dtype, buf)

    def full(self, shape, dtype: str, fill) -> DevArray:
        return self.empty(shape, dtype).fill(fill)

    def release(self, arr: DevArray):
        """Hand one array's block back (no-op if it is not a live block of this pool)."""
        for i, (cls, buf) in enumerate(self._used):
            if buf is arr._owner:
                self._free.setdefault(cls, []).append(buf)
                del self._used[i]
                return

    def release_all(self):
        for cls, buf in self._used:
            self._free.setdefault(cls, []).append(buf)
        self._used = []

    def close(self):
        self.release_all()
        for lst in self._free.values():
            for b in lst:
                b.free()
        self._free = {}


# ------------------------------------------------------------------------------------------------------------------
# the exchange: RCCL behind the C-ABI
def new_unique_id() -> bytes:
    """pvs_comm_unique_id (rank 0): 128 bytes to hand to every rank by any host channel."""
    from . import _ffi
    buf = C.create_string_buffer(128)
    _ffi.check(_ffi.lib().pvs_comm_unique_id(buf))
    return buf.raw


def exchange_unique_id(rank: int, world: int, addr: str, port: int, timeout: float = 300.0) -> bytes:
    """A minimal host channel for the 128-byte id when nothing else is at hand: rank 0 listens on (addr, port) and serves
    world - 1 connections; the others connect (retrying until rank 0 is up)."""
    if world == 1:
        return new_unique_id()
    if rank == 0:
        uid = new_unique_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(struct.pack("<I", len(uid)) + uid)
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                head = b""
                while len(head) < 4:
                    chunk = s.recv(4 - len(head))
                    if not chunk:                      # rank 0 closed early: retry until the deadline, never spin
                        raise ConnectionError("short read (header)")
                    head += chunk
                n, = struct.unpack("<I", head)
                data = b""
                while len(data) < n:
                    chunk = s.recv(n - len(data))
                    if not chunk:
                        raise ConnectionError("short read")
                    data += chunk
                return data
        except (ConnectionError, OSError):
            if time.time() > deadline:
                raise
            time.sleep(0.1)


class RcclComm:
    """One rank of the exchange, bound to a pvsim.Context (collectives are enqueued on that context's stream)."""

    def __init__(self, ctx, world: int, rank: int, unique_id: bytes):
        from . import _ffi
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pvs_comm_init(ctx.handle, self.world, self.rank, C.c_char_p(unique_id), C.byref(h)))
        self.handle = h

    @staticmethod
    def library() -> str:
        from . import _ffi
        return (_ffi.lib().pvs_comm_library() or b"").decode()

    def all_gather(self, send, recv, nbytes: int | None = None):
        """recv[r] = rank r's `send` block; arrays with data_ptr() (and nbytes unless given) or raw pointers + nbytes."""
        from . import _ffi
        n = int(nbytes if nbytes is not None else _nbytes(send))
        _ffi.check(_ffi.lib().pvs_allgather_dev(self.handle, _ffi.ptr(_dptr(send)), _ffi.ptr(_dptr(recv)), n))

    def all_to_all(self, out, inp, nbytes_per_rank: int | None = None):
        from . import _ffi
        n = int(nbytes_per_rank if nbytes_per_rank is not None else _nbytes(inp) // self.world)
        _ffi.check(_ffi.lib().pvs_alltoall_dev(self.handle, _ffi.ptr(_dptr(inp)), _ffi.ptr(_dptr(out)), n))

    def send_recv(self, ops):
        """ops: list of (peer, send_ptr | None, send_bytes, recv_ptr | None, recv_bytes) -- one batched point-to-point group."""
        from . import _ffi
        n = len(ops)
        if not n:
            return
        peers = (C.c_int * n)(*[int(o[0]) for o in ops])
        sp = (C.c_void_p * n)(*[C.c_void_p(int(o[1]) if o[1] else None) for o in ops])
        sb = (C.c_size_t * n)(*[int(o[2]) for o in ops])
        rp = (C.c_void_p * n)(*[C.c_void_p(int(o[3]) if o[3] else None) for o in ops])
        rb = (C.c_size_t * n)(*[int(o[4]) for o in ops])
        _ffi.check(_ffi.lib().pvs_sendrecv_dev(self.handle, n, peers, sp, sb, rp, rb))

    def max_over_ranks(self, values) -> np.ndarray:
        from . import _ffi
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()
        _ffi.check(_ffi.lib().pvs_allreduce_max_f64(self.handle, _ffi.ptr(v), v.shape[0]))
        return v

    def barrier(self):
        from . import _ffi
        _ffi.check(_ffi.lib().pvs_comm_barrier(self.handle))

    def close(self):
        if self.handle is not None:
            from . import _ffi
            if self.ctx.handle is not None:        # pvs_comm_destroy reads the context (device, stream): never after its close
                _ffi.lib().pvs_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dptr(a) -> int:
    return int(a.data_ptr()) if hasattr(a, "data_ptr") else int(a)


def _nbytes(a) -> int:
    if hasattr(a, "nbytes"):
        return int(a.nbytes)
    return int(a.numel() * a.element_size())     # torch tensors


def gather_blocks(enc_loc, inv_loc, comm, empty):
    """All-gather of the (block, L) encodings and (block,) inverse norms -> (world*block, L), (world*block,).
    `enc_loc` must already be padded to the common block size (padding rows are ignored by retrieve_sharded).
    comm: RcclComm-like (all_gather(send, recv)); empty(shape, like) allocates a result array."""
    world = comm.world
    enc_all = empty((world * enc_loc.shape[0], enc_loc.shape[1]), enc_loc)
    inv_all = empty((world * inv_loc.shape[0],), inv_loc)
    comm.all_gather(enc_loc, enc_all)
    comm.all_gather(inv_loc, inv_all)
    return enc_all, inv_all


def mask_padding(inv_loc, n_loc: int) -> None:
    """Mark the padding rows of a rank's block (rows >= n_loc) with a NaN inverse norm IN PLACE, before the
    all-gather: their scores become NaN, which the select kernel ranks below every number, so they can never
    displace a real image.  Because every block has the same size B and only trailing rows are padding, the row
    position in the gathered (world*B, L) array IS the true global image index."""
    if n_loc < inv_loc.shape[0]:
        tail = inv_loc[n_loc:]
        if isinstance(tail, DevArray):
            tail.fill(float("nan"))
        else:
            tail[...] = float("nan")


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


def retrieve_traveling_queries(q_loc, inv_q_loc, db_loc, inv_db_loc, n_db_loc: int, db_offset: int, rank: int, world: int,
                               k: int, comm, all_to_all: Callable, score_block: Callable, merge: Callable, new_array: Callable,
                               idx, val, gather_begin: Callable | None = None, gather_arrived: Callable | None = None, bufs=None):
    """A bounded block of queries per rank against the SHARDED database, without moving the database (eval.py:13-46: a few
    query images against a large index).  The ranks all-gather their query blocks (world x nqB rows -- small), every rank ranks
    ALL queries against its own database block with true global indices, returns each owner its k candidates per query in one
    all-to-all and the owner merges the `world` lists it receives.  The result equals ranking the owner's queries against the
    whole gathered database: a row's score does not depend on the panel it was computed in, and all lists are ordered
    (score desc, index asc).

    q_loc (nqB, L) / inv_q_loc (nqB,): this rank's query block, the same nqB on every rank (pad with any rows; the owner ignores
    the lists of its padding).  db_loc / inv_db_loc: this rank's database block, n_db_loc real rows whose global indices start
    at db_offset.  comm.all_gather(send, recv); all_to_all(out, inp) exchanges equal slabs (RcclComm.all_to_all);
    score_block(q, n_q, db, n_db, inv_q, inv_db, k, col_offset, merge, idx, val); merge(idx_lists, val_lists, n_lists, nq, k,
    idx, val) (DeviceOps.merge); new_array(shape, dtype, fill).  idx / val: (>= nqB, k) outputs.
    gather_begin() / gather_arrived(): stream-ordering hooks around the all-gather (the rank's own queries are scored from the
    local copy between the two, so the gather runs under that launch).  bufs: the tuple a previous call of the same shape
    returned (its six work arrays are used again instead of new ones)."""
    nqB, L = q_loc.shape[0], q_loc.shape[1]
    if bufs is not None:
        q_all, invq_all, part_i, part_v, recv_i, recv_v = bufs
    else:
        q_all = new_array((world * nqB, L), q_loc.dtype, None)
        invq_all = new_array((world * nqB,), inv_q_loc.dtype, None)
        part_i = new_array((world * nqB, k), "int64", -1)
        part_v = new_array((world * nqB, k), "float32", float("-inf"))
        recv_i = new_array((world * nqB, k), "int64", -1)
        recv_v = new_array((world * nqB, k), "float32", float("-inf"))
    if gather_begin is not None:
        gather_begin()
    comm.all_gather(inv_q_loc, invq_all)
    comm.all_gather(q_loc, q_all)
    if n_db_loc > 0:
        score_block(q_loc, nqB, db_loc, n_db_loc, inv_q_loc, inv_db_loc, k, db_offset, False, part_i[rank * nqB:], part_v[rank * nqB:])
    if gather_arrived is not None:
        gather_arrived()
    for r in range(world):
        if r == rank or n_db_loc <= 0:
            continue
        score_block(q_all[r * nqB:], nqB, db_loc, n_db_loc, invq_all[r * nqB:], inv_db_loc, k, db_offset, False,
                    part_i[r * nqB:], part_v[r * nqB:])
    all_to_all(recv_i.reshape(-1), part_i.reshape(-1))      # slab r of part_* goes to rank r; slab s of recv_* comes from rank s
    all_to_all(recv_v.reshape(-1), part_v.reshape(-1))
    merge(recv_i.reshape(world, nqB, k), recv_v.reshape(world, nqB, k), world, nqB, k, idx, val)
    return q_all, invq_all, part_i, part_v, recv_i, recv_v   # (the caller may hand them back to its pool)


def device_score_block(ctx):
    """score_block for device arrays: one pvs_cosine_topk_dev call (GEMM panel + select, running-list merge)."""

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
        """same_stream=True: the context was created on the stream the caller's arrays / collectives use, so no host
        synchronisation is needed between this object's launches and the caller's."""
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
    the rank's local copy, then before_cross() must make the gathered rows of the OTHER ranks visible (e.g. make the compute
    stream wait for the all-gather) before the cross blocks are scored.

    enc_all (world*B, L) / inv_all (world*B,): the gathered encodings and inverse norms (padding rows unused).
    ops: DeviceOps-like.  all_to_all(out, inp): exchange of equal (B*k)-element slabs between ranks (RcclComm.all_to_all).
    new_tensor(shape, dtype, fill): allocator, dtype "int64" or "float32" (DevicePool.full).
    Returns (idx (n_loc, k) int64, val (n_loc, k) float32)."""
    st = symmetric_local(enc_all, inv_all, n_total, rank, world, k, ops, new_tensor, own=own, before_cross=before_cross)
    all_to_all(st["m_idx"][:world].reshape(-1), st["s_idx"].reshape(-1))
    all_to_all(st["m_val"][:world].reshape(-1), st["s_val"].reshape(-1))
    return symmetric_finish(st, ops, new_tensor)


def symmetric_local(enc_all, inv_all, n_total: int, rank: int, world: int, k: int, ops, new_tensor, own=None,
                    before_cross=None) -> dict:
    """Phase 1 (no communication of its own): score this rank's block pairs; returns the send / merge buffers.
    own / before_cross: see retrieve_symmetric."""
    lo, hi, B = shard_range(n_total, world, rank)
    n_r = hi - lo
    P = world
    # slot p < P: list computed by rank p for my queries (received); slot P: what I computed for myself
    m_idx = new_tensor((P + 1, B, k), "int64", -1)
    m_val = new_tensor((P + 1, B, k), "float32", float("-inf"))
    s_idx = new_tensor((P, B, k), "int64", -1)           # slot p: list I computed for rank p's queries
    s_val = new_tensor((P, B, k), "float32", float("-inf"))
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
            panel = new_tensor((n_r, span), "float32", 0.0)
            panel_t = new_tensor((span, n_r), "float32", 0.0)
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
                panel = new_tensor((n_a, w), "float32", 0.0)
                panel_t = new_tensor((w, n_a), "float32", 0.0)
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
    out_i = new_tensor((st["B"], st["k"]), "int64", -1)
    out_v = new_tensor((st["B"], st["k"]), "float32", float("-inf"))
    if st["n_r"] > 0:
        ops.merge(st["m_idx"], st["m_val"], st["P"] + 1, st["B"], st["k"], out_i, out_v)
        ops.sync()
    return out_i[:st["n_r"]], out_v[:st["n_r"]]


class ShardedVLADIndex:
    """Encode this rank's images on its GPU, exchange once, answer all-vs-all top-k for the local block.  torch-free:
    descriptors arrive as host arrays (or device pointers), the exchange is an RcclComm (None = single GPU)."""

    def __init__(self, ctx, codebook, n_total: int, comm=None):
        self.ctx, self.cb, self.n_total, self.comm = ctx, codebook, n_total, comm
        self.world = comm.world if comm is not None else 1
        self.rank = comm.rank if comm is not None else 0
        self.lo, self.hi, self.block = shard_range(n_total, self.world, self.rank)
        self.pool = DevicePool(ctx)

    def encode_local(self, d_desc: int, kind: int, d_offsets: int, total_desc: int, power=1.0, norm_order=2, epsilon=1e-9):
        """d_desc / d_offsets: device pointers of this rank's packed descriptors and CSR offsets."""
        n_loc = self.hi - self.lo
        L = self.cb.K * self.cb.D
        # a new encode starts a new cycle: the blocks of the previous encode / exchange / topk go back to the pool (same
        # stream, so their reuse is ordered behind the work that still reads them) -- repeated cycles do not grow device memory
        self.pool.release_all()
        self.enc_loc = self.pool.full((self.block, L), "float32", 0.0)
        self.inv_loc = self.pool.full((self.block,), "float32", 1.0)
        self.ctx.vlad_encode_dev(self.cb, d_desc, kind, d_offsets, n_loc, total_desc, self.enc_loc.ptr, power, norm_order,
                                 epsilon, d_inv_norm=self.inv_loc.ptr)
        if self.world > 1:
            mask_padding(self.inv_loc, n_loc)
        return self.enc_loc

    def _to_comm(self):
        """The communicator may live on a context (= stream) of its own: its stream waits for this index's work ..."""
        if self.comm is not None and self.comm.ctx is not self.ctx:
            self.comm.ctx.wait_for(self.ctx)

    def _from_comm(self):
        """... and this index's stream for what has been queued on the communicator's."""
        if self.comm is not None and self.comm.ctx is not self.ctx:
            self.ctx.wait_for(self.comm.ctx)

    def exchange(self):
        if self.world > 1:
            self._to_comm()
            self.enc_all, self.inv_all = gather_blocks(self.enc_loc, self.inv_loc, self.comm,
                                                       lambda shape, like: self.pool.empty(shape, like.dtype))
            self._from_comm()
        else:
            self.enc_all, self.inv_all = self.enc_loc, self.inv_loc

    def search(self, q, inv_q, k: int):
        """This rank's block of queries (DevArray (nqB, L) float32 + (nqB,) 1/norm; the same nqB on every rank) against the whole
        sharded index WITHOUT the exchange(): the query blocks travel, the encoded blocks stay (retrieve_traveling_queries).
        -> (idx (nqB, k) int64 global image indices, val (nqB, k) float32), the same lists as topk() would give these rows."""
        nqB = q.shape[0]
        for b in getattr(self, "_search_bufs", ()):
            self.pool.release(b)
        idx = self.pool.full((max(nqB, 1), k), "int64", -1)
        val = self.pool.full((max(nqB, 1), k), "float32", float("-inf"))
        n_loc = self.hi - self.lo
        if self.comm is None:
            self.ctx.cosine_topk_dev(q.data_ptr(), nqB, self.enc_loc.data_ptr(), n_loc, q.shape[1], inv_q.data_ptr(),
                                     self.inv_loc.data_ptr(), k, 0, False, idx.data_ptr(), val.data_ptr())
            self._search_bufs = (idx, val)
        else:
            inv_db = self.inv_loc       # padding rows carry NaN (mask_padding in encode_local) and only n_loc rows are scored anyway
            ops = DeviceOps(self.ctx, same_stream=True)

            def new_array(shape, dtype, fill):
                return self.pool.empty(shape, dtype) if fill is None else self.pool.full(shape, dtype, fill)

            def a2a(out, inp):
                self._to_comm()
                self.comm.all_to_all(out, inp)
                self._from_comm()

            tmp = retrieve_traveling_queries(q, inv_q, self.enc_loc, inv_db, n_loc, self.lo, self.rank, self.world, k, self.comm,
                                             a2a, device_score_block(self.ctx), ops.merge, new_array, idx, val,
                                             gather_begin=self._to_comm, gather_arrived=self._from_comm)
            self._search_bufs = (idx, val) + tuple(tmp)
        self.ctx.sync()
        return idx[:nqB].numpy(), val[:nqB].numpy()

    def topk(self, k: int):
        n_loc = self.hi - self.lo
        for b in getattr(self, "_lists", ()):          # the lists of the previous query on this index
            self.pool.release(b)
        idx = self.pool.full((max(n_loc, 1), k), "int64", -1)
        val = self.pool.full((max(n_loc, 1), k), "float32", float("-inf"))
        self._lists = (idx, val)
        retrieve_sharded(self.enc_loc, self.inv_loc, self.enc_all, self.inv_all, self.n_total, self.rank, self.world,
                         k, device_score_block(self.ctx), idx, val)
        self.ctx.sync()
        return idx[:n_loc].numpy(), val[:n_loc].numpy()
