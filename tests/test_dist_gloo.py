"""world_size-2 gloo test (CPU) of the multi-GPU host logic: contiguous image sharding, padded all-gather of
the encoding blocks, per-block ranking with true global indices merged through a running top-k list.
The GPU scorer is replaced by an oracle-based one (tests may use the oracle); the collective is real (gloo)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


class _GlooComm:
    """torch.distributed (gloo) behind the call contract of pvsim.distributed.RcclComm -- CPU tensors."""

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def all_gather(self, send, recv):
        dist.all_gather_into_tensor(recv, send.contiguous())

    def all_to_all(self, out, inp):
        dist.all_to_all_single(out, inp)


def _empty_like(shape, like):
    return torch.empty(shape, dtype=like.dtype)


def _new_tensor(shape, dtype, fill):
    return torch.full(shape, fill, dtype=getattr(torch, dtype))


def _oracle_score_block(q, n_q, db, n_db, inv_q, inv_db, k, col_offset, merge, idx, val):
    import pvsim_oracle as orc
    s = orc.cosine_similarity(q[:n_q].numpy(), db[:n_db].numpy())
    s = np.where(np.isnan(inv_db[:n_db].numpy())[None, :], -np.inf, s)   # padding rows (NaN inverse norm) rank last
    cand_idx = np.tile(np.arange(col_offset, col_offset + n_db), (n_q, 1))
    cand_val = s
    if merge:
        keep = idx[:n_q].numpy() >= 0
        cand_idx = np.concatenate([np.where(keep, idx[:n_q].numpy(), -1), cand_idx], axis=1)
        cand_val = np.concatenate([np.where(keep, val[:n_q].numpy(), -np.inf), cand_val], axis=1)
    order = np.lexsort((cand_idx, -cand_val), axis=1)[:, :k]          # (score desc, index asc)
    ti = np.take_along_axis(cand_idx, order, 1)
    tv = np.take_along_axis(cand_val, order, 1).astype(np.float32)
    kk = ti.shape[1]
    idx[:n_q, :kk] = torch.from_numpy(ti)
    val[:n_q, :kk] = torch.from_numpy(tv)


def _worker(rank, world, port, n_total, k, out_dir):
    for p in (os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
        sys.path.insert(0, p)
    from pvsim import distributed as pd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    enc = rng.normal(size=(n_total, 48)).astype(np.float32)           # the whole corpus (same on both ranks)
    enc[5] = enc[2]                                                   # a tie across/within blocks
    lo, hi, block = pd.shard_range(n_total, world, rank)
    enc_loc = torch.zeros((block, 48), dtype=torch.float32)
    enc_loc[: hi - lo] = torch.from_numpy(enc[lo:hi])
    inv_loc = torch.ones((block,), dtype=torch.float32)
    pd.mask_padding(inv_loc, hi - lo)
    enc_all, inv_all = pd.gather_blocks(enc_loc, inv_loc, _GlooComm(), _empty_like)
    assert enc_all.shape == (world * block, 48)
    idx = torch.full((block, k), -1, dtype=torch.int64)
    val = torch.full((block, k), float("-inf"), dtype=torch.float32)
    n_loc = pd.retrieve_sharded(enc_loc, inv_loc, enc_all, inv_all, n_total, rank, world, k, _oracle_score_block,
                                idx, val)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=idx[:n_loc].numpy(), val=val[:n_loc].numpy(), lo=lo, hi=hi,
             enc=enc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [11, 16])
def test_sharded_retrieval_two_ranks_gloo(tmp_path, n_total):
    import pvsim_oracle as orc
    from pvsim import distributed as pd
    world, k = 2, 4
    port = 29500 + (os.getpid() % 2000) + n_total
    mp.spawn(_worker, args=(world, port, n_total, k, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    enc = parts[0]["enc"]
    ridx, rval = orc.topk(orc.cosine_similarity(enc, enc), k)          # single-process answer
    got_idx = np.concatenate([p["idx"] for p in parts])
    got_val = np.concatenate([p["val"] for p in parts])
    assert [int(p["lo"]) for p in parts] == [pd.shard_range(n_total, world, r)[0] for r in range(world)]
    assert np.array_equal(got_idx, ridx)
    np.testing.assert_allclose(got_val, rval, atol=1e-6)


def test_shard_range_covers_everything():
    from pvsim.distributed import shard_range
    for n in (0, 1, 7, 8, 8189, 1_000_000):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= blk for lo, hi, blk in spans)


# ----------------------------------------------------------------------------------------- symmetric pair scheme
class _CpuOps:
    """CPU stand-ins (NumPy / oracle) for pvsim.distributed.DeviceOps -- same call contract."""

    @staticmethod
    def _select(scores, cand_idx, k, merge, idx, val):
        n_q = scores.shape[0]
        ci, cv = cand_idx, scores
        if merge:
            keep = idx[:n_q].numpy() >= 0
            ci = np.concatenate([np.where(keep, idx[:n_q].numpy(), -1), ci], axis=1)
            cv = np.concatenate([np.where(keep, val[:n_q].numpy(), -np.inf), cv], axis=1)
        order = np.lexsort((ci, -cv), axis=1)[:, :k]
        ti, tv = np.take_along_axis(ci, order, 1), np.take_along_axis(cv, order, 1).astype(np.float32)
        ti = np.where(np.isneginf(tv), -1, ti)
        idx[:n_q, :ti.shape[1]] = torch.from_numpy(ti)
        val[:n_q, :tv.shape[1]] = torch.from_numpy(tv)

    def sym_topk(self, q, n, inv, k, col_offset, idx, val):
        import pvsim_oracle as orc
        s = orc.cosine_similarity(q[:n].numpy(), q[:n].numpy())
        self._select(s, np.tile(np.arange(col_offset, col_offset + n), (n, 1)), k, False, idx, val)

    def dual(self, a, m, b, n, inv_a, inv_b, panel, panel_t):
        import pvsim_oracle as orc
        s = orc.cosine_similarity(a[:m].numpy(), b[:n].numpy())
        panel[:m, :n] = torch.from_numpy(s)
        panel_t[:n, :m] = torch.from_numpy(np.ascontiguousarray(s.T))

    def topk(self, scores, nq, ncols, k, col_offset, merge, idx, val):
        s = scores.reshape(-1)[: nq * ncols].reshape(nq, ncols).numpy()
        self._select(s, np.tile(np.arange(col_offset, col_offset + ncols), (nq, 1)), k, merge, idx, val)

    def merge(self, idx_lists, val_lists, n_lists, nq, k, idx, val):
        ci = idx_lists.numpy().transpose(1, 0, 2).reshape(nq, n_lists * k)
        cv = np.where(ci >= 0, val_lists.numpy().transpose(1, 0, 2).reshape(nq, n_lists * k), -np.inf)
        self._select(cv, ci, k, False, idx, val)

    def sync(self):
        pass


def _sym_worker(rank, world, port, n_total, k, out_dir):
    for p in (os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
        sys.path.insert(0, p)
    from pvsim import distributed as pd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(321)
    enc = rng.normal(size=(n_total, 40)).astype(np.float32)
    lo, hi, block = pd.shard_range(n_total, world, rank)
    enc_loc = torch.zeros((block, 40), dtype=torch.float32)
    enc_loc[: hi - lo] = torch.from_numpy(enc[lo:hi])
    inv_loc = torch.ones((block,), dtype=torch.float32)
    comm = _GlooComm()
    enc_all, inv_all = pd.gather_blocks(enc_loc, inv_loc, comm, _empty_like)     # real all_gather_into_tensor
    idx, val = pd.retrieve_symmetric(enc_all, inv_all, n_total, rank, world, k, _CpuOps(), comm.all_to_all,   # real all-to-all
                                     _new_tensor)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), idx=idx.numpy(), val=val.numpy(), enc=enc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 21), (3, 20), (4, 30)])
def test_symmetric_pair_scheme_gloo(tmp_path, world, n_total):
    """The symmetric block-pair scheme with REAL collectives (gloo all-gather + all-to-all) and CPU stand-in kernels:
    every pair scored once, lists exchanged, merged -> equals the single-process ranking."""
    import pvsim_oracle as orc
    k = 4
    port = 29900 + (os.getpid() % 1500) + world * 7 + n_total
    mp.spawn(_sym_worker, args=(world, port, n_total, k, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"s{r}.npz") for r in range(world)]
    enc = parts[0]["enc"]
    ridx, rval = orc.topk(orc.cosine_similarity(enc, enc), k)
    assert np.array_equal(np.concatenate([p["idx"] for p in parts]), ridx)
    np.testing.assert_allclose(np.concatenate([p["val"] for p in parts]), rval, atol=1e-6)


def _travel_worker(rank, world, port, n_total, nqB, k, out_dir):
    for p in (os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
        sys.path.insert(0, p)
    from pvsim import distributed as pd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(77)
    enc = rng.normal(size=(n_total, 36)).astype(np.float32)
    enc[7] = enc[1]                                                   # a tie between rows of different blocks
    lo, hi, block = pd.shard_range(n_total, world, rank)
    db = torch.zeros((block, 36), dtype=torch.float32)
    db[: hi - lo] = torch.from_numpy(enc[lo:hi])
    inv_db = torch.ones((block,), dtype=torch.float32)
    # this rank's queries: the first rows of its block (fewer than nqB on a short last block: the rest is padding)
    nq = min(nqB, hi - lo)
    q = torch.zeros((nqB, 36), dtype=torch.float32)
    q[:nq] = db[:nq]
    inv_q = torch.ones((nqB,), dtype=torch.float32)
    comm = _GlooComm()
    idx = torch.full((nqB, k), -1, dtype=torch.int64)
    val = torch.full((nqB, k), float("-inf"), dtype=torch.float32)

    def new_array(shape, dtype, fill):
        dt = getattr(torch, dtype) if isinstance(dtype, str) else dtype
        return torch.empty(shape, dtype=dt) if fill is None else torch.full(shape, fill, dtype=dt)

    bufs = pd.retrieve_traveling_queries(q, inv_q, db, inv_db, hi - lo, lo, rank, world, k, comm, comm.all_to_all, _oracle_score_block,
                                         _CpuOps().merge, new_array, idx, val)
    first = (idx.clone(), val.clone())
    pd.retrieve_traveling_queries(q, inv_q, db, inv_db, hi - lo, lo, rank, world, k, comm, comm.all_to_all, _oracle_score_block,
                                  _CpuOps().merge, new_array, idx, val, bufs=bufs)              # work arrays used again
    assert torch.equal(first[0], idx) and torch.equal(first[1], val)
    np.savez(os.path.join(out_dir, f"t{rank}.npz"), idx=idx[:nq].numpy(), val=val[:nq].numpy(), lo=lo, nq=nq, enc=enc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,nqB", [(2, 23, 4), (3, 20, 3), (4, 9, 3)])
def test_bounded_query_blocks_travel_the_database_stays_gloo(tmp_path, world, n_total, nqB):
    """retrieve_traveling_queries with REAL collectives (gloo all-gather of the query blocks, all-to-all of the candidate lists)
    and CPU stand-in kernels: every rank's lists equal the single-process ranking of its queries against the whole corpus
    ((4, 9, 3): a short last block and a rank without any rows)."""
    import pvsim_oracle as orc
    k = 4
    port = 27100 + (os.getpid() % 1500) + world * 11 + n_total
    mp.spawn(_travel_worker, args=(world, port, n_total, nqB, k, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"t{r}.npz") for r in range(world)]
    enc = parts[0]["enc"]
    ridx, rval = orc.topk(orc.cosine_similarity(enc, enc), k)
    for p_ in parts:
        lo, nq = int(p_["lo"]), int(p_["nq"])
        assert np.array_equal(p_["idx"], ridx[lo:lo + nq])
        np.testing.assert_allclose(p_["val"], rval[lo:lo + nq], atol=1e-6)


def test_devarray_views_and_unique_id_bootstrap():
    """DevArray's first-axis views (the only tensor protocol the retrieval logic needs) and the socket hand-off of the
    128-byte communicator id between two processes (no GPU: the id is replaced by a byte pattern)."""
    import socket
    import threading
    from pvsim import distributed as pd
    a = pd.DevArray(None, 1000, (3, 4, 5), "int64")
    assert a[1].data_ptr() == 1000 + 160 and a[1].shape == (4, 5)
    assert a[1:].shape == (2, 4, 5) and a[1:][0].data_ptr() == a[1].data_ptr()
    assert a[2][1:].data_ptr() == 1000 + 2 * 160 + 40 and a[2][1:].shape == (3, 5)
    assert a[:2].reshape(-1).shape == (40,) and a.reshape(-1).nbytes == 480
    assert a[1:1].shape == (0, 4, 5)
    with pytest.raises(IndexError):
        a[3]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    uid = bytes(range(128))
    orig = pd.new_unique_id
    pd.new_unique_id = lambda: uid
    got = {}
    try:
        ts = [threading.Thread(target=lambda r=r: got.__setitem__(r, pd.exchange_unique_id(r, 3, "127.0.0.1", port, 30.0))) for r in range(3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
    finally:
        pd.new_unique_id = orig
    assert got == {0: uid, 1: uid, 2: uid}


@pytest.mark.parametrize("n_total,world,n_chunks", [(1000000, 8, 4), (20000, 2, 4), (30011, 3, 3), (7, 4, 4), (5, 8, 2), (1030, 1, 4), (100, 3, 1)])
def test_exchange_chunks_tile_the_corpus(n_total, world, n_chunks):
    """bench.py's chunked fp16 exchange (chunk c scored while chunk c + 1 travels): the (rank, chunk) pieces cover every image
    index exactly once, in the order of the blocks, and a rank's own pieces are exactly its block."""
    from pvsim import distributed as pd
    chunks, pieces = pd.exchange_chunks(n_total, world, n_chunks)
    _, _, block = pd.shard_range(n_total, world, 0)
    assert chunks[0][0] == 0 and chunks[-1][1] == block and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    seen = np.zeros(n_total, np.int32)
    for ci, (c0, c1) in enumerate(chunks):
        for r in range(world):
            g0, nv = pieces[ci][r]
            assert g0 == r * block + c0 and 0 <= nv <= c1 - c0
            seen[g0:g0 + nv] += 1
            lo, hi, _ = pd.shard_range(n_total, world, r)
            assert nv == 0 or (lo <= g0 and g0 + nv <= hi)
    assert (seen == 1).all()
