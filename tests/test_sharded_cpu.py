"""Host logic of the multi-GPU path (verticut_amd/sharded.py) on CPU: world_size-2 gloo ranks, the GPU backend
replaced by an oracle-backed double (the C-ABI calls themselves are covered by the -m gpu tests).  What is
checked here: shard ranges / id bases, the all-gather layout the merge kernel expects, identical results on
every rank, equality with the unsharded database."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from verticut_amd.sharded import ShardedSearch, shard_range

INF = np.uint64(0xFFFFFFFFFFFFFFFF)


class OracleBackend:
    """CPU stand-in with the GpuBackend interface (tests only)."""

    def __init__(self, vo, bits, id_base):
        self.vo, self.bits, self.id_base, self.codes = vo, bits, id_base, None

    def add_synthetic(self, n, seed, kind=0, n_centres=0, max_flips=0):
        self.codes = self.vo.gen_codes(n, self.bits, seed, kind, n_centres, max_flips, first_id=self.id_base)

    def local_topk(self, queries, k, out, counts, mode):
        q = queries.numpy()
        for i in range(q.shape[0]):
            r = self.vo.linear_knn(self.codes, q[i], k, id_base=self.id_base)
            row = np.full(k, INF, dtype=np.uint64)
            row[: len(r)] = r
            out[i] = torch.from_numpy(row.view(np.int64))
            counts[i] = len(r)

    def merge(self, gathered, world, nq, k, out, counts):
        g = gathered.numpy().view(np.uint64)          # [world][nq][k], the layout vc_merge_topk_dev reads
        assert g.shape == (world, nq, k)
        for i in range(nq):
            allv = np.sort(g[:, i, :].reshape(-1))
            out[i] = torch.from_numpy(allv[:k].copy().view(np.int64))
            counts[i] = int((allv[:k] != INF).sum())

    def close(self):
        pass


def _worker(rank, world, port, total_n, bits, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vc_oracle as vo
    lo, hi = shard_range(total_n, rank, world)
    ss = ShardedSearch(bits, total_n, rank=rank, world=world, backend=OracleBackend(vo, bits, lo))
    ss.add_synthetic(34, kind=1, n_centres=20, max_flips=6)
    full = vo.gen_codes(total_n, bits, 34, 1, 20, 6)
    q = full[[3, total_n // 2, total_n - 1]].copy()
    q[:, 0] ^= 0x11
    out, cnt = ss.search(torch.from_numpy(q), k)
    got = out.numpy().view(np.uint64).copy()
    exp = np.stack([vo.linear_knn(full, q[i], k) for i in range(len(q))])
    ret[rank] = bool(np.array_equal(got, exp) and np.all(cnt.numpy() == k))
    dist.destroy_process_group()


def _worker_bucketed(rank, world, port, total_n, bits, k, ret):
    """bucket = 3, five batches: one full bucket exchanged by the third call, a partial one by flush()"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vc_oracle as vo
    lo, hi = shard_range(total_n, rank, world)
    ss = ShardedSearch(bits, total_n, rank=rank, world=world, backend=OracleBackend(vo, bits, lo), bucket=3)
    ss.add_synthetic(34, kind=1, n_centres=20, max_flips=6)
    full = vo.gen_codes(total_n, bits, 34, 1, 20, 6)
    rng = np.random.default_rng(5)
    ok = True
    pending = []

    def check_pending():
        nonlocal ok
        for q, h in pending:
            out, cnt = h.get()
            exp = np.stack([vo.linear_knn(full, q[i], k) for i in range(len(q))])
            ok = ok and np.array_equal(out.numpy().view(np.uint64), exp) and bool(np.all(cnt.numpy() == k))
        pending.clear()

    from verticut_amd.sharded import PendingResult
    ready_seen = []
    for b in range(5):
        q = full[rng.integers(0, total_n, size=4)].copy()
        q[:, b] ^= 0x21
        h = ss.search(torch.from_numpy(q), k)
        ok = ok and isinstance(h, PendingResult)
        pending.append((q, h))
        ready_seen.append(h.ready)
        if (b + 1) % 3 == 0:
            ok = ok and all(x.ready for _, x in pending)       # the third call exchanged the bucket
            check_pending()
    ok = ok and ready_seen == [False, False, True, False, False]
    first_of_partial = pending[0][1]
    out, cnt = first_of_partial                                # unpacking a handle = get(): exchanges the partial bucket on demand
    ok = ok and first_of_partial.ready and pending[1][1].ready
    ss.flush()                                                 # nothing left to do
    check_pending()
    # a handle that outlives two further buckets refuses to hand out recycled buffers
    q = full[:4].copy()
    stale = ss.search(torch.from_numpy(q), k)
    for _ in range(2 * 3):
        ss.search(torch.from_numpy(q), k)
    try:
        stale.get()
        ok = False
    except RuntimeError:
        pass
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_cover_and_balance():
    for total in (1, 7, 1000, 10 ** 9, 4 * 10 ** 9):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, i, world) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_search_equals_unsharded(oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), 6001, 128, 25, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_two_rank_gloo_bucketed_exchange(oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bucketed, args=(world, _free_port(), 5003, 128, 12, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_single_rank_needs_no_collective(oracle):
    from oracle import vc_oracle as vo
    ss = ShardedSearch(128, 500, rank=0, world=1, backend=OracleBackend(vo, 128, 0))
    ss.add_synthetic(1)
    q = vo.gen_codes(2, 128, 9)
    out, cnt = ss.search(torch.from_numpy(q), 10)
    full = vo.gen_codes(500, 128, 1)
    assert np.array_equal(out.numpy().view(np.uint64)[0], vo.linear_knn(full, q[0], 10))
