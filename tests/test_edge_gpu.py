"""Edge cases of the C ABI on the GPU: empty / tiny databases, extreme k, many query tiles, argument errors."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INF = np.uint64(0xFFFFFFFFFFFFFFFF)


def test_empty_database(vc):
    q = np.zeros((3, 16), dtype=np.uint8)
    with vc.Engine(128, capacity=100, n_tables=4) as e:
        out, cnt = e.search_knn(q, 10)
        assert np.all(cnt == 0) and np.all(out == INF)
        e.build_index()
        out, cnt, st = e.search_knn(q, 10, mode=vc.MODE_MIH_EXACT, with_stats=True)
        assert np.all(cnt == 0) and all(s.radius == 32 for s in st)     # loop runs to the last shell (search_worker.cc:170)
        assert all(len(r) == 0 for r in e.search_radius(q, 5))
        assert all(len(r) == 0 for r in e.search_radius(q, 5, mode=vc.MODE_MIH_EXACT))
        assert e.get_code(0) is None and e.get_bucket(0, 0) is None


def test_single_item_and_k1(vc, oracle):
    code = np.arange(16, dtype=np.uint8)[None, :]
    with vc.Engine(128, capacity=1, n_tables=4, id_base=77) as e:
        e.add_codes(code)
        e.build_index()
        q = code.copy()
        q[0, 0] ^= 0x7
        for mode in (vc.MODE_LINEAR, vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
            out, cnt = e.search_knn(q, 1, mode=mode)
            assert cnt[0] == 1 and int(out[0, 0]) == (3 << 32 | 77)


def test_max_k_and_k_larger_than_db(vc, oracle):
    n, bits = 20000, 64
    codes = oracle.gen_codes(n, bits, 3)
    q = codes[:2].copy()
    with vc.Engine(bits, capacity=n) as e:
        e.add_synthetic(n, seed=3)
        out, cnt = e.search_knn(q, 8192)                       # VC_MAX_K
        exp = oracle.linear_knn(codes, q[0], 8192)
        assert cnt[0] == 8192 and np.array_equal(out[0], exp)
        with pytest.raises(vc.VcError) as ei:
            e.search_knn(q, 8193)
        assert ei.value.code == vc.VC_ERR_INVALID
    with vc.Engine(bits, capacity=5000) as e:
        e.add_codes(codes[:5000])
        out, cnt = e.search_knn(q, 8192)                       # k > N: everything, ascending, INF padded
        exp = oracle.linear_knn(codes[:5000], q[1], 8192)
        assert cnt[1] == 5000 and np.array_equal(out[1, :5000], exp) and np.all(out[1, 5000:] == INF)


def test_many_query_tiles(vc, oracle):
    n, bits, k, nq = 30000, 128, 7, 203                        # 203 queries = 6 full tiles of 32 + 11
    rng = np.random.default_rng(1)
    codes = oracle.gen_codes(n, bits, 9, kind=1, n_centres=40, max_flips=9)
    q = codes[rng.integers(0, n, size=nq)].copy()
    q[:, 4] ^= 0x21
    with vc.Engine(bits, capacity=n, n_tables=4) as e:
        e.add_synthetic(n, seed=9, kind=1, n_centres=40, max_flips=9)
        out, cnt = e.search_knn(q, k)
        for i in range(nq):
            assert np.array_equal(out[i], oracle.linear_knn(codes, q[i], k)), i
        e.build_index()
        mout, mcnt = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT)   # > one MIH tile? (256) no; but all queries at once
        assert np.array_equal(mout >> np.uint64(32), out >> np.uint64(32))


def test_more_queries_than_one_mih_tile(vc, oracle):
    n, bits, k, nq = 20000, 64, 5, 300                          # MIH tiles hold 256 queries
    rng = np.random.default_rng(2)
    codes = oracle.gen_codes(n, bits, 4, kind=1, n_centres=30, max_flips=4)
    q = codes[rng.integers(0, n, size=nq)].copy()
    with vc.Engine(bits, capacity=n, n_tables=4) as e:
        e.add_codes(codes)
        e.build_index()
        lin, _ = e.search_knn(q, k)
        mih, _, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        assert np.array_equal(mih >> np.uint64(32), lin >> np.uint64(32))
        assert all(s.n_results == k for s in st)
        res = e.search_radius(q, 6, mode=vc.MODE_MIH_EXACT)
        ref = e.search_radius(q, 6, mode=vc.MODE_LINEAR)
        assert all(np.array_equal(a, b) for a, b in zip(res, ref))


def test_argument_errors(vc):
    L = vc.load_library()

    def create(**kw):
        cfg = vc.VcConfig(abi_version=vc.VC_ABI_VERSION, bits=128, n_tables=4, capacity=10, device=-1)
        for k_, v in kw.items():
            setattr(cfg, k_, v)
        h = C.c_void_p()
        rc = L.vc_create(C.byref(cfg), C.byref(h))
        if rc == 0:
            L.vc_destroy(h)
        return rc

    assert create() == vc.VC_OK
    assert create(abi_version=99) == vc.VC_ERR_INVALID
    assert create(bits=100) == vc.VC_ERR_INVALID
    assert create(n_tables=3) == vc.VC_ERR_INVALID            # 16 bytes not divisible (search_worker.cc:75 assert)
    assert create(bits=256, n_tables=4) == vc.VC_ERR_INVALID  # 64-bit substrings: binaryToInt stops at 32
    assert create(capacity=0) == vc.VC_ERR_INVALID
    assert create(capacity=1 << 32, id_base=1) == vc.VC_ERR_INVALID   # ids are uint32
    assert create(device=99) == vc.VC_ERR_NO_DEVICE
    with vc.Engine(128, capacity=10) as e:                   # linear-only engine
        with pytest.raises(vc.VcError) as ei:
            e.build_index()
        assert ei.value.code == vc.VC_ERR_STATE
        with pytest.raises(vc.VcError):
            e.get_bucket(0, 0)
        with pytest.raises(vc.VcError) as ei:
            e.add_synthetic(11, seed=1)
        assert ei.value.code == vc.VC_ERR_CAPACITY
        with pytest.raises(vc.VcError):
            e.search_knn(np.zeros((1, 16), np.uint8), 0)
        with pytest.raises(ValueError):
            e.search_knn(np.zeros((1, 8), np.uint8), 1)


def test_device_api_matches_host_api(vc, oracle):
    import torch
    n, bits, k = 50000, 128, 20
    codes = oracle.gen_codes(n, bits, 6)
    q = codes[:9].copy()
    q[:, 9] ^= 0x55
    with vc.Engine(bits, capacity=n) as e:
        e.add_synthetic(n, seed=6)
        host, hcnt = e.search_knn(q, k)
        dq = torch.from_numpy(q).cuda()
        out = torch.empty((9, k), dtype=torch.int64, device="cuda")
        cnt = torch.empty((9,), dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        e.search_knn_dev(dq.data_ptr(), 9, k, out.data_ptr(), cnt.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), host) and np.array_equal(cnt.cpu().numpy(), hcnt)
        t = e.timing()
        assert t.calls == 2 and t.scan_launches == 2
    # VC_FLAG_LEAN_TIMING: same results, only the verify launches are bracketed by events
    with vc.Engine(bits, capacity=n, flags=vc.FLAG_LEAN_TIMING) as e:
        e.add_synthetic(n, seed=6)
        lean, lcnt = e.search_knn(q, k)
        e.search_knn_dev(dq.data_ptr(), 9, k, out.data_ptr(), cnt.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert np.array_equal(lean, host) and np.array_equal(out.cpu().numpy().view(np.uint64), host)
        t = e.timing()
        assert t.calls == 0 and t.total_ms == 0 and t.scan_launches == 2 and t.scan_ms > 0


def test_device_api_is_ordered_on_the_callers_stream(vc):
    """results of vc_search_knn_dev must be visible to later work on the SAME stream without any host sync
    (the multi-GPU exchange reads them from the stream right away), on the null stream and on a side stream."""
    import torch
    n, bits, k = 60_000_000, 128, 50                     # ~0.25 ms of scan: long enough for a race to show
    rng = np.random.default_rng(8)
    q1 = rng.integers(0, 256, size=(8, 16), dtype=np.uint8)
    q2 = rng.integers(0, 256, size=(8, 16), dtype=np.uint8)
    with vc.Engine(bits, capacity=n, query_tile=8) as e:
        e.add_synthetic(n, seed=34)
        ref1, _ = e.search_knn(q1, k)
        ref2, _ = e.search_knn(q2, k)
        d1, d2 = torch.from_numpy(q1).cuda(), torch.from_numpy(q2).cuda()
        out = torch.zeros((8, k), dtype=torch.int64, device="cuda")
        cnt = torch.zeros((8,), dtype=torch.int32, device="cuda")
        for stream in (torch.cuda.current_stream(), torch.cuda.Stream()):
            with torch.cuda.stream(stream):
                for dq, ref in ((d1, ref1), (d2, ref2), (d1, ref1)):
                    e.search_knn_dev(dq.data_ptr(), 8, k, out.data_ptr(), cnt.data_ptr(), stream=stream.cuda_stream)
                    snap = out.clone()                   # torch op on the same stream, no synchronisation in between
                    stream.synchronize()
                    assert np.array_equal(snap.cpu().numpy().view(np.uint64), ref)
        # a host-API call right after an un-synchronised device-API call must not trample the shared work buffers
        e.search_knn_dev(d2.data_ptr(), 8, k, out.data_ptr(), cnt.data_ptr(), stream=None)
        again, _ = e.search_knn(q1, k)
        torch.cuda.synchronize()
        assert np.array_equal(again, ref1) and np.array_equal(out.cpu().numpy().view(np.uint64), ref2)


def test_results_do_not_depend_on_the_column_stride(vc, oracle, monkeypatch):
    """vc_create may pad the column stride (it times candidate strides for big databases): every path -- ingest,
    synthetic fill, get_code, linear and MIH search, radius search, code file save -- must honour whatever stride was
    chosen.  VC_STRIDE_FORCE picks odd ones for a small database."""
    n, bits, k = 70_000, 128, 25
    codes = oracle.gen_codes(n, bits, 4, kind=1, n_centres=150, max_flips=7)
    rng = np.random.default_rng(2)
    q = codes[rng.integers(0, n, size=10)].copy()
    q[:, 2] ^= 0x09
    ref = None
    for force in (None, "81920", "131072", "212992"):
        if force is None:
            monkeypatch.delenv("VC_STRIDE_FORCE", raising=False)
        else:
            monkeypatch.setenv("VC_STRIDE_FORCE", force)
        with vc.Engine(bits, capacity=n, n_tables=4, query_tile=4) as e:
            if force == "131072":
                e.add_synthetic(n, seed=4, kind=1, n_centres=150, max_flips=7)
            else:
                e.add_codes(codes)
            assert np.array_equal(e.get_code(n - 1), codes[n - 1]) and np.array_equal(e.get_code(12345), codes[12345])
            lin, lcnt = e.search_knn(q, k)
            e.build_index()
            mih, mcnt = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT)
            rad = e.search_radius(q[:3], 9, mode=vc.MODE_MIH_EXACT)
            got = (lin.copy(), lcnt.copy(), mih.copy(), mcnt.copy(), [r.copy() for r in rad])
        if ref is None:
            ref = got
            for i in range(len(q)):
                assert np.array_equal(lin[i, : lcnt[i]], oracle.linear_knn(codes, q[i], k))
        else:
            assert all(np.array_equal(a, b) for a, b in zip(got[:4], ref[:4]))
            assert all(np.array_equal(a, b) for a, b in zip(got[4], ref[4]))

@pytest.mark.parametrize("bits", [64, 128, 512])
def test_cache_resident_prefix_does_not_change_results(vc, oracle, monkeypatch, bits):
    """The verify kernel reads a prefix of the database with plain loads (it stays in the Infinity Cache from pass to
    pass) and the rest non-temporal (VC_SCAN_RESIDENT_MB, default 240): same bytes, same results wherever the boundary
    falls -- no prefix, a boundary inside the database (1 MB of 3.2-25 MB), everything resident -- and on repeated passes."""
    n, k = 400_000, 50
    codes = oracle.gen_codes(n, bits, 9, kind=1, n_centres=300, max_flips=9)
    rng = np.random.default_rng(bits)
    q = codes[rng.integers(0, n, size=9)].copy()
    q[:, 1] ^= 0x21
    exp = [oracle.linear_knn(codes, q[i], k) for i in range(len(q))]
    for mb in ("0", "1", None):
        if mb is None:
            monkeypatch.delenv("VC_SCAN_RESIDENT_MB", raising=False)
        else:
            monkeypatch.setenv("VC_SCAN_RESIDENT_MB", mb)
        with vc.Engine(bits, capacity=n, query_tile=8) as e:
            e.add_codes(codes)
            for _ in range(2):
                got, cnt = e.search_knn(q, k)
                for i in range(len(q)):
                    assert np.array_equal(got[i, : cnt[i]], exp[i]), (mb, i)


@pytest.mark.parametrize("n_lists,nq,k", [(1, 3, 7), (2, 8, 100), (8, 8, 100), (8, 4096, 100), (16, 5, 1), (5, 9, 333), (16, 3, 1000), (3, 2, 8192)])
def test_merge_of_sorted_lists_equals_numpy(vc, n_lists, nq, k):
    """vc_merge_topk_dev (gather_vectors' consumer + the master heap, mpi_coordinator.cc:34-69, search_worker.cc:179-199):
    ascending INF-padded lists -> the k smallest, ascending, INF-padded, valid counts -- on the LDS merge kernel (up to
    6144 entries per query) and on the general select kernel behind it; partly filled and empty lists, a value that occurs
    in several lists (overlapping shards are not the engine's case, but the entry point takes any lists)"""
    import torch
    rng = np.random.default_rng(n_lists * 1000 + k)
    INF = np.uint64(0xFFFFFFFFFFFFFFFF)
    lists = np.full((n_lists, nq, k), INF, dtype=np.uint64)
    for g in range(n_lists):
        for q in range(nq if nq <= 64 else 64):
            fill = int(rng.integers(0, k + 1)) if (g + q) % 3 else k
            vals = (rng.integers(0, 128, size=fill).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 32, size=fill).astype(np.uint64)
            vals = np.unique(vals)
            lists[g, q, : len(vals)] = vals
    if nq > 64:
        lists[:, 64:] = lists[:, :1]                               # the same query many times: every list equal = duplicates across lists
    if n_lists > 1 and k > 1:
        lists[1, 0, 0] = lists[0, 0, 0]                             # a duplicate across two lists, at the front
        lists[1, 0] = np.sort(lists[1, 0])
    d = torch.from_numpy(lists.view(np.int64)).cuda()
    out = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    vc.merge_topk_dev(d.data_ptr(), n_lists, nq, k, out.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    got, c = out.cpu().numpy().view(np.uint64), cnt.cpu().numpy()
    for q in range(nq):
        allv = np.sort(lists[:, q].reshape(-1))
        exp = allv[:k]
        assert np.array_equal(got[q], exp), (q, n_lists, k)
        assert c[q] == int((exp != INF).sum())


@pytest.mark.parametrize("nq,k", [(6, 100), (700, 100), (3000, 100), (1500, 400)])
def test_host_pointer_calls_pageable_and_page_locked(vc, oracle, nq, k):
    """vc_search_knn's three ways home for the rows (SearchWorker::find's caller holds host memory, search_worker.cc:65-89): one
    pinned staging buffer (<= 512 KB), two pinned chunks with the DMA of one overlapping the host copy of the other (pageable
    destination), one DMA (page-locked destination).  All equal the device-resident call and the oracle's linear scan."""
    import torch
    n, bits = 40_000, 128
    rng = np.random.default_rng(nq + k)
    codes = oracle.gen_codes(n, bits, 5, kind=1, n_centres=200, max_flips=8)
    q = codes[rng.integers(0, n, size=nq)].copy()
    q[:, 0] ^= rng.integers(0, 4, size=nq, dtype=np.uint8)
    with vc.Engine(bits, capacity=n, n_tables=4) as e:
        e.add_codes(codes)
        e.build_index()
        for mode in (vc.MODE_LINEAR, vc.MODE_MIH_EXACT):
            ref, rcnt = e.search_knn(q, k, mode=mode)                                   # fresh pageable arrays
            out = np.zeros((nq, k), dtype=np.uint64)
            cnt = np.zeros(nq, dtype=np.uint32)
            e.search_knn(q, k, mode=mode, out=out, counts=cnt)                          # caller's pageable arrays
            assert np.array_equal(out, ref) and np.array_equal(cnt, rcnt)
            pout = torch.zeros((nq, k), dtype=torch.int64).pin_memory().numpy().view(np.uint64)
            pcnt = torch.zeros((nq,), dtype=torch.int32).pin_memory().numpy().view(np.uint32)
            pq = torch.from_numpy(q).pin_memory().numpy()
            e.search_knn(pq, k, mode=mode, out=pout, counts=pcnt)                       # page-locked: straight DMA
            assert np.array_equal(pout, ref) and np.array_equal(pcnt, rcnt)
            d_q = torch.from_numpy(q).cuda()
            d_out = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            d_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
            e.search_knn_dev(d_q.data_ptr(), nq, k, d_out.data_ptr(), d_cnt.data_ptr(), mode=mode)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy().view(np.uint64), ref)
        lin = e.search_knn(q[:6], k)[0]
        for i in range(6):
            assert np.array_equal(lin[i], oracle.linear_knn(codes, q[i], k))
        with pytest.raises(ValueError):
            e.search_knn(q, k, out=np.zeros((nq, k + 1), dtype=np.uint64))
