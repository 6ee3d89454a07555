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
        held.append((r, sb.search(torch.from_numpy(qq).cuda(), k)[0]))
        if r == 2 or r == 4:
            if r == 4:
                sb.flush()
            torch.cuda.synchronize()
            for rr, o in held:
                ok = ok and bool(np.array_equal(o.cpu().numpy().view(np.uint64), np.roll(exp, rr, axis=0)))
            held = []
    ret[rank] = ok
    ss.close()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_equals_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, 200001, 128, 100, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}
