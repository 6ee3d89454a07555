"""Multi-rank path on the one GPU of the test box: two processes share cuda:0, each owns half of the id range
(real engines, real merge kernel); the exchange goes over gloo because RCCL wants one device per rank.  The
N-GPU RCCL run itself is the driver's; this pins everything around the collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, total_n, bits, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vc_oracle as vo
    from verticut_amd import engine as vc
    from verticut_amd.sharded import ShardedSearch
    ss = ShardedSearch(bits, total_n, rank=rank, world=world, device=0, n_tables=4)
    ss.add_synthetic(34, kind=vc.SYNTH_CLUSTERED, n_centres=300, max_flips=10)
    full = vo.gen_codes(total_n, bits, 34, 1, 300, 10)
    rng = np.random.default_rng(5)
    q = full[rng.integers(0, total_n, size=9)].copy()
    q[:, 3] ^= 0x24
    dq = torch.from_numpy(q).cuda()
    out, cnt = ss.search(dq, k)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    exp = np.stack([vo.linear_knn(full, q[i], k) for i in range(len(q))])
    ok = bool(np.array_equal(got, exp) and np.all(cnt.cpu().numpy() == k))
    # MIH on shards: every shard's exact top-k merged == exact top-k of the union (distances; ties may differ)
    ss.build_index()
    out2, _ = ss.search(dq, k, mode=vc.MODE_MIH_EXACT)
    torch.cuda.synchronize()
    got2 = out2.cpu().numpy().view(np.uint64)
    ok = ok and bool(np.array_equal(got2 >> np.uint64(32), exp >> np.uint64(32)))
    # pipelined exchange (side stream, two buffer sets): five batches in flight, every result still exact
    sp = ShardedSearch(bits, total_n, rank=rank, world=world, device=0, pipelined=True, backend=ss.backend)
    outs = []
    for r in range(5):
        qq = np.roll(q, r, axis=0).copy()
        o, _ = sp.search(torch.from_numpy(qq).cuda(), k)
        outs.append((r, o))                  # a buffer set is reused two batches later, by design
        if r < 3:
            sp.flush()
            torch.cuda.synchronize()
            ok = ok and bool(np.array_equal(o.cpu().numpy().view(np.uint64), np.roll(exp, r, axis=0)))
    sp.flush()
    torch.cuda.synchronize()
    for r in (3, 4):                         # the last two batches sit in the two buffer sets
        ok = ok and bool(np.array_equal(outs[r][1].cpu().numpy().view(np.uint64), np.roll(exp, r, axis=0)))
    # bucketed exchange: three batches per all-gather + merge, the fourth and fifth completed by flush()
    sb = ShardedSearch(bits, total_n, rank=rank, world=world, device=0, bucket=3, backend=ss.backend)
    held = []
    for r in range(5):
        qq = np.roll(q, r, axis=0).copy()
        held.append((r, sb.search(torch.from_numpy(qq).cuda(), k)))      # PendingResult handles
        if r == 2 or r == 4:
            ok = ok and all(h.ready for _, h in held) == (r == 2)          # the third call exchanged its bucket; 4 and 5 wait
            if r == 4:
                sb.flush()
            torch.cuda.synchronize()
            for rr, h in held:
                o, c = h.get()
                ok = ok and bool(np.array_equal(o.cpu().numpy().view(np.uint64), np.roll(exp, rr, axis=0))) and bool(np.all(c.cpu().numpy() == k))
            held = []
    ret[rank] = ok
    ss.close()
    dist.destroy_process_group()


def _worker_overflow(rank, world, port, total_n, bits, k, cand_cap, ret):
    """duplicate-heavy shards whose candidate rings overflow: the asynchronous device path must hand exact per-shard
    top-k to the exchange (round-1 hole: the overflow marker was dropped and the merged row was silently wrong)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vc_oracle as vo
    from verticut_amd.sharded import ShardedSearch
    full = vo.gen_codes(total_n, bits, 900, 1, 7, 0)          # 7 distinct codes x ~57 K copies
    ss = ShardedSearch(bits, total_n, rank=rank, world=world, device=0, cand_cap=cand_cap, query_tile=4)
    ss.add_codes_global(full)
    rng = np.random.default_rng(3)
    q = full[rng.integers(0, total_n, size=9)].copy()
    q[2, 1] ^= 0x40
    q[8] = rng.integers(0, 256, size=bits // 8, dtype=np.uint8)
    ok = True
    for bucket in (1, 3):
        sb = ShardedSearch(bits, total_n, rank=rank, world=world, device=0, bucket=bucket, backend=ss.backend)
        out, cnt = sb.search(torch.from_numpy(q).cuda(), k)
        sb.flush()
        torch.cuda.synchronize()
        exp = np.stack([vo.linear_knn(full, q[i], k) for i in range(len(q))])
        ok = ok and bool(np.array_equal(out.cpu().numpy().view(np.uint64), exp)) and bool(np.all(cnt.cpu().numpy() == k))
        ok = ok and sb.unrecovered() == 0
    ret[rank] = ok
    ss.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("cand_cap,k", [(4, 1), (256, 100)])
def test_two_ranks_duplicate_heavy_shards_overflow(cand_cap, k):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_overflow, args=(2, port, 400_000, 256, k, cand_cap, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_two_ranks_one_gpu_equals_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, 200001, 128, 100, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}
