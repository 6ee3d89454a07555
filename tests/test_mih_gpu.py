"""GPU parity of the MIH path (search_worker.cc, build_hash_tables.cc, bitmap.cc) against the oracle.

Contract: per query the sorted distance array equals the oracle's, the id set below
the k-th distance equals the oracle's, ids at the k-th distance are genuine; radius / n_sub_reads /
n_local_reads equal the oracle's.  On top of that the engine's own canonical rule is checked exactly:
result == the k smallest (dist, id) among the items whose minimum substring distance is <= radius.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SH = np.uint64(32)
INF = np.uint64(0xFFFFFFFFFFFFFFFF)


def _near_queries(codes, nq, rng, max_flips):
    q = codes[rng.integers(0, codes.shape[0], size=nq)].copy()
    nb = codes.shape[1]
    for i in range(nq):
        for _ in range(rng.integers(0, max_flips + 1)):
            b = rng.integers(0, nb * 8)
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    return q


def _reach_min_subdist(vo, codes, q, m, signext):
    """min over tables of the substring distance; with sign-extended keys a table only reaches items whose
    substring top bit equals the query's (Pilaf/image_tools.h:13)."""
    sub = vo.np_sub_distances(codes, q, m).astype(np.int64)
    if signext:
        nlb = codes.shape[1] // m
        x = np.bitwise_xor(codes, q[None, :]).reshape(codes.shape[0], m, nlb)
        top_differs = (x[:, :, nlb - 1] & 0x80) != 0
        sub[top_differs] = 10 ** 6
    return sub.min(axis=1)


def _canonical_mih(vo, codes, q, m, k, radius, signext, id_base=0):
    seen = _reach_min_subdist(vo, codes, q, m, signext) <= radius
    d = vo.np_distances(codes, q)
    ids = np.arange(codes.shape[0], dtype=np.uint64) + np.uint64(id_base)
    packed = np.sort(vo.pack(d[seen], ids[seen]))
    return packed[:k], int(seen.sum())


def _check_contract(got, oracle_res):
    """distance multiset + id set below the k-th distance."""
    o = np.sort(oracle_res)
    assert len(got) == len(o)
    assert np.array_equal(got >> SH, o >> SH)
    if len(o):
        dk = o[-1] >> SH
        assert set(got[(got >> SH) < dk].tolist()) == set(o[(o >> SH) < dk].tolist())


CONFIGS = [
    # bits, m, flags-name, n, centres, flips, k
    (128, 4, "", 60000, 300, 10, 100),       # the reference's native shape: 4 x 32-bit substrings
    (128, 4, "", 60000, 300, 10, 1),
    (64, 4, "", 40000, 200, 5, 50),          # 16-bit substrings, masked keys (exact)
    (64, 4, "signext", 40000, 200, 5, 50),   # 16-bit substrings, binaryToInt sign-extension quirk reproduced
    (64, 2, "", 40000, 200, 6, 20),          # 2 tables: stop multiplier min(m,4) = 2
    (64, 2, "literal4", 40000, 200, 6, 20),  # the reference's literal-4 stop rule (may stop early; parity only)
    (256, 8, "", 30000, 150, 14, 100),       # 8 x 32-bit substrings
    (64, 8, "", 20000, 100, 4, 10),          # 8-bit substrings
]


@pytest.mark.parametrize("bits,m,fl,n,centres,flips,k", CONFIGS)
def test_mih_exact_parity(vc, oracle, bits, m, fl, n, centres, flips, k):
    rng = np.random.default_rng(bits + 31 * m + k)
    flags = {"": 0, "signext": vc.FLAG_REF_SIGNEXT_KEYS, "literal4": vc.FLAG_REF_STOP_LITERAL4}[fl]
    signext = fl == "signext"
    stop_mult = 4 if fl == "literal4" else min(m, 4)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=centres, max_flips=flips)
    mo = oracle.MihOracle(codes, m, key_mode=0 if signext else 1)
    q = _near_queries(codes, 12, rng, flips // 2)
    with vc.Engine(bits, capacity=n, n_tables=m, flags=flags) as e:
        e.add_synthetic(n, seed=34, kind=1, n_centres=centres, max_flips=flips)
        e.build_index()
        got, cnt, stats = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        lin, _ = e.search_knn(q, k, mode=vc.MODE_LINEAR)
        for i in range(q.shape[0]):
            ores, ost = mo.find(q[i], k, approximate=False, use_bitmap=False, stop_mult=stop_mult)
            g = got[i, : cnt[i]]
            _check_contract(g, ores)
            assert stats[i].radius == ost.radius, (i, stats[i].radius, ost.radius)
            assert stats[i].n_sub_reads == ost.n_sub_reads
            assert stats[i].n_local_reads == 0 and stats[i].n_main_reads == 0
            assert stats[i].n_candidates == ost.n_distinct
            exp, _ = _canonical_mih(oracle, codes, q[i], m, k, ost.radius, signext)
            assert np.array_equal(g, exp)
            if fl == "":  # exact configurations agree with the linear scan on distances
                assert np.array_equal(g >> SH, lin[i, : cnt[i]] >> SH)


@pytest.mark.parametrize("bits,m,n,k", [(128, 4, 60000, 10), (64, 4, 40000, 10)])
def test_mih_approximate_parity(vc, oracle, bits, m, n, k):
    rng = np.random.default_rng(77 + bits)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=100, max_flips=10)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    q = _near_queries(codes, 8, rng, 3)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_synthetic(n, seed=34, kind=1, n_centres=100, max_flips=10)
        e.build_index()
        got, cnt, stats = e.search_knn(q, k, mode=vc.MODE_MIH_APPROX, with_stats=True)
        for i in range(q.shape[0]):
            ores, ost = mo.find(q[i], k, approximate=True, stop_mult=4)
            g = got[i, : cnt[i]]
            assert stats[i].radius == ost.radius
            assert stats[i].n_candidates == ost.n_distinct
            assert np.array_equal(g >> SH, np.sort(ores) >> SH)
            exp, _ = _canonical_mih(oracle, codes, q[i], m, k, ost.radius, False)
            assert np.array_equal(g, exp)


def test_mih_bitmap_stats(vc, oracle):
    """with the bitmap attached (search_worker.cc:238-245): n_local_reads = leaves, n_sub_reads = set bits."""
    n, bits, m, k = 50000, 128, 4, 20
    rng = np.random.default_rng(3)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=200, max_flips=8)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    q = _near_queries(codes, 6, rng, 3)
    with vc.Engine(bits, capacity=n, n_tables=m, flags=vc.FLAG_USE_BITMAP) as e:
        e.add_codes(codes)
        e.build_index()
        got, cnt, stats = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        for i in range(q.shape[0]):
            ores, ost = mo.find(q[i], k, use_bitmap=True, stop_mult=4)
            _check_contract(got[i, : cnt[i]], ores)
            assert (stats[i].radius, stats[i].n_sub_reads, stats[i].n_local_reads) == \
                (ost.radius, ost.n_sub_reads, ost.n_local_reads)


@pytest.mark.parametrize("bits,m,signext", [(128, 4, False), (64, 4, False), (64, 4, True), (64, 8, False)])
def test_bucket_and_bitmap_views(vc, oracle, bits, m, signext):
    """HashIndex -> Image_List get (rule a12) and ImageBitmap bits, bucket by bucket."""
    n = 30000
    rng = np.random.default_rng(bits + m)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=40, max_flips=3)
    mo = oracle.MihOracle(codes, m, key_mode=0 if signext else 1, id_base=500)
    flags = vc.FLAG_REF_SIGNEXT_KEYS if signext else 0
    with vc.Engine(bits, capacity=n, n_tables=m, flags=flags, id_base=500) as e:
        e.add_codes(codes)
        e.build_index()
        for t in range(m):
            present = [mo.key(codes[i], t) for i in rng.integers(0, n, size=6)]
            absent = [int(x) for x in rng.integers(0, 1 << 32, size=4, dtype=np.uint64)]
            for idx in present + absent:
                exp = mo.bucket(t, idx)
                res = e.get_bucket(t, idx, cap=1 << 15)
                assert e.bitmap_test(t, idx) == (1 if len(exp) else 0)
                if len(exp) == 0:
                    assert res is None
                    continue
                ids, bcodes, total = res
                assert total == len(exp) and np.array_equal(ids, exp)
                assert np.array_equal(bcodes, codes[exp - 500])
        if bits // m <= 16:  # raw bitmap words as generate_bitmap.cc writes them
            words = e.bitmap_read(0, 0, (1 << (bits // m)) // 32)
            keys = {mo.key(codes[i], 0) & ((1 << (bits // m)) - 1) for i in range(n)}
            exp_words = np.zeros_like(words)
            for kk in keys:
                oracle.lib().vco_bitmap_set(exp_words.ctypes.data, kk)
            assert np.array_equal(words, exp_words)


@pytest.mark.parametrize("n", [1, 63, 4095, 4096, 4097, 16383, 16384, 16385, 50001])
def test_index_build_sort_is_stable_at_tile_boundaries(vc, oracle, n):
    """The builder's hand-written radix sort (vc_sort.hip: 4096-item sub-tiles, 16384-item blocks): the WHOLE id array
    of a table, bucket by bucket, for record counts around its tile sizes -- 8-bit substrings (one pass, every bucket
    enumerated), 16-bit (two passes) and 32-bit (four passes, sampled)."""
    rng = np.random.default_rng(n)
    for bits, m, tables in ((64, 8, (0, 7)), (64, 4, (1,)), (128, 4, (2,))):
        codes = oracle.gen_codes(n, bits, 77 + n, kind=0)
        mo = oracle.MihOracle(codes, m, key_mode=1, id_base=0)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            for t in tables:
                keys = sorted({mo.key(codes[i], t) for i in range(n)})
                if len(keys) > 300:
                    keys = [keys[i] for i in sorted(set(rng.integers(0, len(keys), size=120).tolist()) | {0, len(keys) - 1})]
                seen = 0
                for key in keys:
                    exp = mo.bucket(t, key)
                    ids, bcodes, total = e.get_bucket(t, key, cap=1 << 16)
                    assert total == len(exp) and np.array_equal(ids, exp), (bits, m, t, key)
                    seen += total
                if bits // m == 8:
                    assert seen == n


def test_mih_overflow_recovery(vc, oracle):
    """a shell whose candidates overflow the ring is re-run with a tightened limit; results unchanged."""
    n, bits, m, k = 80000, 128, 4, 100
    rng = np.random.default_rng(8)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=8, max_flips=2)  # ~10k items per bucket
    q = _near_queries(codes, 5, rng, 1)
    with vc.Engine(bits, capacity=n, n_tables=m, cand_cap=512) as e, vc.Engine(bits, capacity=n, n_tables=m) as big:
        for eng in (e, big):
            eng.add_codes(codes)
            eng.build_index()
        a, ca, sa = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        b, cb, sb = big.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        assert np.array_equal(a, b) and np.array_equal(ca, cb)
        for x, y in zip(sa, sb):
            assert (x.radius, x.n_sub_reads, x.n_candidates) == (y.radius, y.n_sub_reads, y.n_candidates)
        for i in range(q.shape[0]):
            exp, _ = _canonical_mih(oracle, codes, q[i], m, k, sa[i].radius, False)
            assert np.array_equal(a[i, : ca[i]], exp)


@pytest.mark.parametrize("bits,m,radius", [(64, 4, 8), (64, 2, 8), (128, 4, 12)])
def test_radius_search(vc, oracle, bits, m, radius):
    """BASELINE config 2 shape: all neighbours within full distance r, MIH == linear == numpy."""
    n = 50000
    rng = np.random.default_rng(radius + m)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=100, max_flips=6)
    q = _near_queries(codes, 9, rng, 3)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        lin = e.search_radius(q, radius, mode=vc.MODE_LINEAR)
        mih = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT)
        for i in range(q.shape[0]):
            d = oracle.np_distances(codes, q[i])
            ids = np.nonzero(d <= radius)[0]
            exp = np.sort(oracle.pack(d[ids], ids.astype(np.uint64)))
            assert np.array_equal(lin[i], exp)
            assert np.array_equal(mih[i], exp)


@pytest.mark.parametrize("bits,m", [(64, 2), (64, 4), (64, 8), (128, 4)])
def test_radius_search_every_remainder_of_the_pigeonhole_split(vc, oracle, bits, m):
    """Radius search gives table t the substring radius q (t <= a) or q - 1 (t > a) for R = m q + a -- the tables beyond
    a are not searched at all while R < m.  Every R from 0 over a few multiples of m, items at EVERY distance around the
    query (flips spread over the substrings on purpose), MIH == linear == numpy; both the query-kernel path and the
    multi-block shell kernels (VC_MIH_HOST_LOOP=1 is covered by the host-loop test; here the budget decides)."""
    n = 30000
    rng = np.random.default_rng(bits + m)
    base = oracle.gen_codes(1, bits, 5)[0]
    codes = np.tile(base, (n, 1))
    for i in range(n):                                  # 0..2m+3 flips at random positions: every split of R over the substrings
        for b in rng.choice(bits, size=int(rng.integers(0, 2 * m + 4)), replace=False):
            codes[i, b // 8] ^= np.uint8(1 << (b % 8))
    q = base[None, :].copy()
    d = oracle.np_distances(codes, q[0])
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        for radius in range(0, 2 * m + 3):
            ids = np.nonzero(d <= radius)[0]
            exp = np.sort(oracle.pack(d[ids], ids.astype(np.uint64)))
            mih = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT, cap_per_query=1 << 15)
            assert np.array_equal(mih[0], exp), (radius, len(mih[0]), len(exp))


def test_mih_needs_index_and_rejects_stale(vc):
    with vc.Engine(128, capacity=1000, n_tables=4) as e:
        e.add_synthetic(500, seed=1, kind=vc.SYNTH_CLUSTERED, n_centres=5, max_flips=2)
        q = e.get_code(0)[None, :]
        with pytest.raises(vc.VcError) as ei:
            e.search_knn(q, 5, mode=vc.MODE_MIH_EXACT)
        assert ei.value.code == vc.VC_ERR_STATE
        e.build_index()
        e.search_knn(q, 5, mode=vc.MODE_MIH_EXACT)
        e.add_synthetic(100, seed=2, kind=vc.SYNTH_CLUSTERED, n_centres=5, max_flips=2)  # invalidates the index
        with pytest.raises(vc.VcError):
            e.search_knn(q, 5, mode=vc.MODE_MIH_EXACT)


def test_bucket_order_code_copies_change_nothing(vc, oracle, monkeypatch):
    """<= 16-bit substrings verify their buckets from a bucket-order copy of the codes (VcTableView::bcodes): same
    results, statistics and radius search as through the id gather (VC_MIH_BCODES=0)."""
    n, bits, m, k = 120_000, 64, 4, 30
    rng = np.random.default_rng(21)
    codes = oracle.gen_codes(n, bits, 8, kind=1, n_centres=400, max_flips=6)
    q = _near_queries(codes, 16, rng, 3)
    got = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("VC_MIH_BCODES", flag)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            res, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
            rad = e.search_radius(q[:6], 7, mode=vc.MODE_MIH_EXACT)
            got[flag] = (res.copy(), cnt.copy(), [(s.radius, s.n_sub_reads, s.n_candidates) for s in st], [r.copy() for r in rad])
    a, b = got["0"], got["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    mo = oracle.MihOracle(codes, m, key_mode=1)
    ores, ost = mo.find(q[0], k, stop_mult=4)
    _check_contract(b[0][0, : b[1][0]], ores)


@pytest.mark.parametrize("bits,m,k", [(128, 4, 100), (64, 4, 50), (64, 2, 20)])
def test_host_loop_path_equals_the_query_kernel(vc, oracle, monkeypatch, bits, m, k):
    """VC_MIH_HOST_LOOP=1 runs every shell through the multi-block kernels (one launch sequence per shell, the round-1
    loop, still the path of the shells beyond the query kernel's budget): same rows, counts, statistics, radius search."""
    n = 50000
    rng = np.random.default_rng(bits + m)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=250, max_flips=8)
    q = _near_queries(codes, 14, rng, 4)
    got = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("VC_MIH_HOST_LOOP", flag)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            res, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
            ares, acnt, ast = e.search_knn(q, 5, mode=vc.MODE_MIH_APPROX, with_stats=True)
            rad = e.search_radius(q[:5], 9, mode=vc.MODE_MIH_EXACT)
            got[flag] = (res, cnt, [(s.radius, s.n_sub_reads, s.n_candidates) for s in st], ares, acnt,
                         [(s.radius, s.n_candidates) for s in ast], rad)
    a, b = got["0"], got["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and a[5] == b[5]
    assert all(np.array_equal(x, y) for x, y in zip(a[6], b[6]))
    mo = oracle.MihOracle(codes, m, key_mode=1)
    for i in range(3):
        ores, ost = mo.find(q[i], k, stop_mult=min(m, 4))
        _check_contract(a[0][i, : a[1][i]], ores)
        assert a[2][i][:2] == (ost.radius, ost.n_sub_reads)


def test_queries_that_outlive_the_query_kernel_continue_in_the_multi_block_shells(vc, oracle):
    """uniform random 128-bit codes: the k-th neighbour is ~40 bits away, so the radius loop runs to shell 8-10, far
    beyond the shells the one-block-per-query kernel covers (0..4): state hand-over to the multi-block kernels, mixed
    with queries that finish in the first shells; radius / n_sub_reads must be the oracle's."""
    n, bits, m, k = 3000, 128, 4, 3
    codes = oracle.gen_codes(n, bits, 77)
    rng = np.random.default_rng(1)
    q = np.stack([codes[5], rng.integers(0, 256, size=16, dtype=np.uint8), codes[100], rng.integers(0, 256, size=16, dtype=np.uint8)])
    q[2, 0] ^= 1
    mo = oracle.MihOracle(codes, m, key_mode=1)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        lin, _ = e.search_knn(q, k)
        assert np.array_equal(got >> SH, lin >> SH)
        for i in (0, 2):                                   # the far queries cost the CPU oracle ~10^8 probes: check the near ones
            ores, ost = mo.find(q[i], k, stop_mult=4)
            _check_contract(got[i, : cnt[i]], ores)
            assert (st[i].radius, st[i].n_sub_reads) == (ost.radius, ost.n_sub_reads)
        assert max(s.radius for s in st) > 4               # the hand-over really happened
        for s in st:                                       # n_sub_reads = every leaf of every shell searched (no bitmap attached)
            assert s.n_sub_reads == sum(__import__("math").comb(32, r) for r in range(s.radius + 1))


@pytest.mark.parametrize("k", [2000, 5000, 8000])
def test_mih_large_k(vc, oracle, k):
    """k = 2000 still runs in the query kernel (top-k + candidates in a 4096-entry LDS buffer); k = 5000 needs the
    8192-entry buffer = 82 KB of dynamic LDS, more than the classic 64 KiB per workgroup (the engine checks the device's
    limit and falls back to the multi-block shells where it does not fit); k = 8000 exceeds what one block keeps in LDS
    and takes the multi-block path from shell 0: all equal the canonical rule and the linear scan."""
    n, bits, m = 60000, 128, 4
    rng = np.random.default_rng(k)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=6, max_flips=6)     # ~10 K items per cluster
    q = _near_queries(codes, 3, rng, 2)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        lin, _ = e.search_knn(q, k)
        assert np.all(cnt == k) and np.array_equal(got >> SH, lin >> SH)
        for i in range(len(q)):
            exp, _ = _canonical_mih(oracle, codes, q[i], m, k, st[i].radius, False)
            assert np.array_equal(got[i], exp)


def test_radius_search_device_api(vc, oracle):
    """vc_search_radius_dev: queries, results and offsets stay in HBM; same rows as the host API and numpy; a too small
    output buffer is reported with the needed size in the offsets; more than one tile of queries."""
    import torch
    n, bits, m, radius = 80000, 64, 2, 8
    rng = np.random.default_rng(4)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=300, max_flips=5)
    q = _near_queries(codes, 4200, rng, 3)                      # two tiles of the query kernel (4096 + 104)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        host = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT, cap_per_query=1 << 12)
        total = sum(len(r) for r in host)
        dq = torch.from_numpy(q).cuda()
        d_off = torch.zeros((len(q) + 1,), dtype=torch.int64, device="cuda")
        d_out = torch.zeros((total + 10,), dtype=torch.int64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for mode in (vc.MODE_MIH_EXACT, vc.MODE_LINEAR):
            rc = e.search_radius_dev(dq.data_ptr(), len(q), radius, d_out.data_ptr(), total + 10, d_off.data_ptr(), mode=mode, stream=s)
            torch.cuda.synchronize()
            assert rc == vc.VC_OK
            off = d_off.cpu().numpy().view(np.uint64)
            res = d_out.cpu().numpy().view(np.uint64)
            assert int(off[-1]) == total
            for i in (0, 1, 500, 4095, 4096, 4199):
                assert np.array_equal(res[int(off[i]):int(off[i + 1])], host[i]), (mode, i)
        for i in (0, 7, 4150):                                   # host rows against numpy
            d = oracle.np_distances(codes, q[i])
            ids = np.nonzero(d <= radius)[0]
            assert np.array_equal(host[i], np.sort(oracle.pack(d[ids], ids.astype(np.uint64))))
        small = torch.zeros((total // 2,), dtype=torch.int64, device="cuda")
        rc = e.search_radius_dev(dq.data_ptr(), len(q), radius, small.data_ptr(), total // 2, d_off.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert rc == vc.VC_ERR_CAPACITY and int(d_off.cpu().numpy().view(np.uint64)[-1]) == total
        t = e.timing()
        assert t.mih_launches >= 4 and t.mih_queries >= 2 * len(q) and t.mih_probes > 0 and t.mih_ms > 0   # 2 tiles x (host + device call)


@pytest.mark.parametrize("bits,m", [(128, 4), (64, 4)])
def test_radius_search_with_huge_neighbourhoods(vc, oracle, bits, m):
    """duplicate-heavy data: a query has 20 000+ neighbours inside the radius -- more than the query kernel keeps in LDS
    (spill to the ring, unsorted), more than the default work ring of 4096 entries holds (the call doubles it and
    repeats) and more than the LDS segment sort takes (8192: bitonic network on the segment in global memory); mixed
    with queries that have a handful.  MIH == linear scan == numpy, ascending, through both APIs' shared path."""
    n = 90000
    rng = np.random.default_rng(bits)
    base = oracle.gen_codes(4, bits, 3)
    codes = base[rng.integers(0, 4, size=n)].copy()                      # ~22 500 copies of each of 4 codes
    flip = rng.integers(0, n, size=n // 3)
    codes[flip, rng.integers(0, bits // 8, size=len(flip))] ^= (1 << rng.integers(0, 8, size=len(flip))).astype(np.uint8)
    codes[:300] = oracle.gen_codes(300, bits, 9)                          # a few loners
    q = np.stack([base[0], codes[5], base[2], codes[100000 % n]])
    q[2, 1] ^= 0x3
    radius = 3
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        mih = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT, cap_per_query=64)      # host buffer too small at first, too
        lin = e.search_radius(q, radius, mode=vc.MODE_LINEAR, cap_per_query=64)
        for i in range(len(q)):
            d = oracle.np_distances(codes, q[i])
            ids = np.nonzero(d <= radius)[0]
            exp = np.sort(oracle.pack(d[ids], ids.astype(np.uint64)))
            assert np.array_equal(mih[i], exp), i
            assert np.array_equal(lin[i], exp), i
        assert max(len(r) for r in mih) > 20000 and min(len(r) for r in mih) < 10


@pytest.mark.parametrize("bits,m", [(512, 16), (512, 64), (256, 8)])
def test_wide_codes_many_tables(vc, oracle, bits, m):
    """512-bit codes: 16 tables of 32-bit substrings (16 occupancy bitmaps, 16 mask sets in the query kernel's LDS) and 64
    tables of 8-bit substrings; exact k-NN statistics against the oracle, radius search against numpy."""
    n, k = 20000, 10
    rng = np.random.default_rng(bits + m)
    codes = oracle.gen_codes(n, bits, 5, kind=1, n_centres=100, max_flips=3 * m // 4 + 2)
    q = _near_queries(codes, 6, rng, 3)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        for i in range(len(q)):
            ores, ost = mo.find(q[i], k, stop_mult=4)
            _check_contract(got[i, : cnt[i]], ores)
            assert (st[i].radius, st[i].n_sub_reads, st[i].n_candidates) == (ost.radius, ost.n_sub_reads, ost.n_distinct)
        radius = m                                          # substring shells 0..1
        rad = e.search_radius(q[:3], radius, mode=vc.MODE_MIH_EXACT)
        for i in range(3):
            d = oracle.np_distances(codes, q[i])
            ids = np.nonzero(d <= radius)[0]
            assert np.array_equal(rad[i], np.sort(oracle.pack(d[ids], ids.astype(np.uint64))))


def test_radius_larger_than_the_probe_budget_falls_back_to_the_scan(vc, oracle):
    """a radius whose substring shells would enumerate more keys than the shard has items (here: every key of a 32-bit
    substring) is answered by the scan: same rows as numpy, in finite time"""
    n, bits, m = 5000, 128, 4
    codes = oracle.gen_codes(n, bits, 8)
    q = codes[:2].copy()
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        for radius in (60, 128):
            rad = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT, cap_per_query=8192)
            for i in range(2):
                d = oracle.np_distances(codes, q[i])
                ids = np.nonzero(d <= radius)[0]
                assert np.array_equal(rad[i], np.sort(oracle.pack(d[ids], ids.astype(np.uint64))))
        assert len(rad[0]) == n                              # radius = bits: everything


@pytest.mark.parametrize("bits,m,fl", [(64, 4, ""), (64, 4, "signext"), (64, 8, ""), (128, 8, "")])
def test_radius_search_through_the_bucket_streaming_kernel(vc, oracle, monkeypatch, bits, m, fl):
    """<= 16-bit substrings whose shells exceed the query kernel's entry budget stream their buckets from the bucket-order
    code copies (mih_bucket_stream_kernel); VC_MIH_STREAM=2 sends small databases through it too.  Every radius around
    a query with neighbours at every distance (all pigeonhole remainders), near-duplicate queries on clustered data,
    duplicate-heavy buckets with 20 000+ neighbours (ring growth + repeat), the sign-extended-key quirk; against numpy
    and the oracle's restatement of search_R_neighbors (search_worker.cc:222-264)."""
    monkeypatch.setenv("VC_MIH_STREAM", "2")
    signext = fl == "signext"
    flags = vc.FLAG_REF_SIGNEXT_KEYS if signext else 0
    n = 30000
    rng = np.random.default_rng(bits + m + len(fl))
    base = oracle.gen_codes(1, bits, 5)[0]
    codes = np.tile(base, (n, 1))
    for i in range(n):
        for b in rng.choice(bits, size=int(rng.integers(0, 2 * m + 4)), replace=False):
            codes[i, b // 8] ^= np.uint8(1 << (b % 8))
    q = base[None, :].copy()
    d = oracle.np_distances(codes, q[0])
    mo = oracle.MihOracle(codes, m, key_mode=0 if signext else 1)
    with vc.Engine(bits, capacity=n, n_tables=m, flags=flags) as e:
        e.add_codes(codes)
        e.build_index()
        for radius in range(0, 2 * m + 3):
            mih = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT, cap_per_query=1 << 15)
            ores, _ = mo.radius(q[0], radius)
            assert np.array_equal(mih[0], ores), (radius, len(mih[0]), len(ores))
            if not signext:                      # masked keys are exact: == brute force
                ids = np.nonzero(d <= radius)[0]
                assert np.array_equal(mih[0], np.sort(oracle.pack(d[ids], ids.astype(np.uint64))))
        t = e.timing()
        assert t.mih_launches > 0 and t.mih_entries > 0 and t.mih_probes > 0      # the streaming kernel is instrumented
    # clustered data, several queries per call, a small batch (the probe list is split over blocks) and a big one
    codes = oracle.gen_codes(60000, bits, 34, kind=1, n_centres=40, max_flips=6)
    with vc.Engine(bits, capacity=len(codes), n_tables=m, flags=flags) as e:
        e.add_codes(codes)
        e.build_index()
        mo = oracle.MihOracle(codes, m, key_mode=0 if signext else 1)
        for nq in (3, 700):
            qq = _near_queries(codes, nq, rng, 3)
            mih = e.search_radius(qq, m + 2, mode=vc.MODE_MIH_EXACT, cap_per_query=4096)
            for i in range(0, nq, max(1, nq // 25)):
                ores, _ = mo.radius(qq[i], m + 2)
                assert np.array_equal(mih[i], ores), (nq, i)


def test_polled_and_synchronised_waits_return_the_same(vc, oracle, monkeypatch):
    """The host learns a k-NN launch's unfinished-query counter and a radius call's total from mapped host memory, polling
    the sequence number the device writes last (VC_MIH_POLL=1, default), or with a read-back copy + hipStreamSynchronize
    (=0): same rows, counts, statistics and radius results, also for queries that continue in the multi-block shells and
    for calls of more than one 4096-query tile."""
    n, bits, m, k = 80000, 128, 4, 20
    rng = np.random.default_rng(9)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=300, max_flips=9)
    q = np.concatenate([_near_queries(codes, 4300, rng, 5), rng.integers(0, 256, size=(3, 16), dtype=np.uint8)])
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VC_MIH_POLL", flag)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            res, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
            res2, cnt2 = e.search_knn(q[:50], k, mode=vc.MODE_MIH_EXACT)
            rad = e.search_radius(q[:4200], 6, mode=vc.MODE_MIH_EXACT)
            rad2 = e.search_radius(q[:7], 6, mode=vc.MODE_MIH_EXACT)
            out[flag] = (res, cnt, [(s.radius, s.n_sub_reads, s.n_candidates) for s in st], res2, cnt2, rad, rad2)
    a, b = out["1"], out["0"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert np.array_equal(a[0][:50], a[3])
    assert len(a[5]) == len(b[5]) and all(np.array_equal(x, y) for x, y in zip(a[5], b[5]))
    assert all(np.array_equal(x, y) for x, y in zip(a[6], b[6])) and all(np.array_equal(x, y) for x, y in zip(a[6], a[5][:7]))
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        lin, lcnt = e.search_knn(q[-3:], k)
        assert np.array_equal(a[0][-3:] >> SH, lin >> SH)      # the uniform queries went through the multi-block shells or the switch
        exp = e.search_radius(q[:7], 6, mode=vc.MODE_LINEAR)
        assert all(np.array_equal(x, y) for x, y in zip(a[6], exp))


def test_work_counters_of_vc_get_timing_are_the_closed_forms(vc, oracle, monkeypatch):
    """vc_get_timing's MIH totals (the algorithmic bytes of bench.py's roofline come from them): probes per query are the closed
    forms of search_worker.cc:222-264 / :170-207 -- radius search: the pigeonhole shells of every table; exact k-NN: m x the
    leaves of shells 0 .. radius of each query -- and the query count is the number of queries, whichever kernel summed them
    (mih_work_reduce_kernel, the radius offsets kernel, the bucket streaming kernel)."""
    from math import comb
    monkeypatch.setenv("VC_MIH_BUDGET", "600000")                 # shells 0..4 inside the query kernel whatever the batch size
    n = 60000
    rng = np.random.default_rng(17)
    # 64-bit, m = 2 (32-bit substrings): R = 8 -> radii 4 and 3
    codes = oracle.gen_codes(n, 64, 34, kind=1, n_centres=300, max_flips=6)
    q = _near_queries(codes, 37, rng, 3)
    with vc.Engine(64, capacity=n, n_tables=2) as e:
        e.add_codes(codes)
        e.build_index()
        e.timing()
        e.search_radius(q, 8, mode=vc.MODE_MIH_EXACT)
        t = e.timing()
        per_query = sum(comb(32, r) for r in range(5)) + sum(comb(32, r) for r in range(4))
        assert per_query == 46938
        assert (t.mih_queries, t.mih_probes) == (len(q), len(q) * per_query) and t.mih_launches == 1
        assert 0 < t.mih_hits <= t.mih_entries
        res, cnt, st = e.search_knn(q, 10, mode=vc.MODE_MIH_EXACT, with_stats=True)
        t = e.timing()
        assert max(s.radius for s in st) <= 4                    # everything finished inside the query kernel
        assert t.mih_queries == len(q) and t.mih_probes == sum(2 * sum(comb(32, r) for r in range(s.radius + 1)) for s in st)
    # 64-bit, m = 4 (16-bit substrings): R = 8 -> radius 2 for table 0, 1 for the others; small buckets: the query kernel
    with vc.Engine(64, capacity=n, n_tables=4) as e:
        e.add_codes(codes)
        e.build_index()
        e.timing()
        e.search_radius(q, 8, mode=vc.MODE_MIH_EXACT)
        t = e.timing()
        per_query = sum(comb(16, r) for r in range(3)) + 3 * sum(comb(16, r) for r in range(2))
        assert per_query == 188
        assert (t.mih_queries, t.mih_probes) == (len(q), len(q) * per_query)


def test_work_counters_of_the_bucket_streaming_kernel(vc, oracle, monkeypatch):
    from math import comb
    monkeypatch.setenv("VC_MIH_STREAM", "2")                      # small databases through mih_bucket_stream_kernel too
    n = 50000
    rng = np.random.default_rng(18)
    codes = oracle.gen_codes(n, 64, 34, kind=1, n_centres=300, max_flips=6)
    q = _near_queries(codes, 21, rng, 3)
    with vc.Engine(64, capacity=n, n_tables=4) as e:
        e.add_codes(codes)
        e.build_index()
        e.timing()
        e.search_radius(q, 8, mode=vc.MODE_MIH_EXACT)
        t = e.timing()
        assert (t.mih_queries, t.mih_probes) == (len(q), len(q) * 188) and t.mih_entries >= t.mih_hits > 0


@pytest.mark.parametrize("bits,m", [(128, 4), (64, 2)])
def test_directory_lines_every_encoding(vc, oracle, monkeypatch, bits, m):
    """Directory lines of the 32-bit tables (VcTableView::lines: occupancy + first entry position + rank + bucket extents of 128
    keys in ONE 64-byte sector; built by default only from 3e8 records on, forced here) against the oracle's
    SearchWorker::find (search_worker.cc:159-264), on data made to hit every encoding of a line: cumulative 16-bit ends
    (<= 16 buckets, some of them long: clusters), 2-bit lengths (dense key ranges whose buckets hold <= 4 entries), the
    fall-back to offsets[rank] (dense ranges with longer buckets), empty lines; exact and approximate mode, statistics,
    bucket views; the same queries with the lines off give the same rows."""
    rng = np.random.default_rng(bits)
    parts = []
    def region(count, lo, span):                       # substrings drawn from [lo, lo + span): density decides the encoding
        return (lo + rng.integers(0, span, size=(count, m))).astype(np.uint32)
    parts.append(region(900, 0, 1 << 10))                                 # ~0.9 entries per key, ~75 buckets per line: 2-bit lengths
    parts.append(region(4000, 1 << 12, 1 << 9))                           # ~8 per key, 128 buckets per line: fall-back
    centres = rng.integers(0, 1 << 32, size=(40, m), dtype=np.uint64)
    pick = rng.integers(0, 40, size=6000)
    parts.append(((centres[pick] ^ rng.integers(0, 8, size=(6000, m)).astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32))   # clusters: <= 8 long buckets per line
    parts.append(rng.integers(0, 1 << 32, size=(5000, m), dtype=np.uint64).astype(np.uint32))                                           # sparse singles
    sub = np.concatenate(parts)
    n_near = sub.shape[0] - 5000                       # queries come from the dense ranges and the clusters (the oracle walks a far
    perm = rng.permutation(sub.shape[0])               # query's shells one key at a time)
    codes = np.ascontiguousarray(sub[perm]).view(np.uint8).reshape(sub.shape[0], bits // 8)
    n, k = codes.shape[0], 15
    near = np.nonzero(perm < n_near)[0]
    q = codes[rng.choice(near, size=24, replace=False)].copy()
    for r in range(len(q)):
        for b in rng.choice(bits, size=int(rng.integers(0, 3)), replace=False):
            q[r, b // 8] ^= np.uint8(1 << (b % 8))
    mo = oracle.MihOracle(codes, m, key_mode=1)
    rows = {}
    for lines in ("1", "0"):
        monkeypatch.setenv("VC_MIH_LINES", lines)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            for mode in (vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
                k = 15 if mode == vc.MODE_MIH_EXACT else 5     # (approximate mode stops at 20 k candidates: a cluster holds ~150 items)
                got, cnt, st = e.search_knn(q, k, mode=mode, with_stats=True)
                rows[(lines, mode)] = got.copy()
                for i in range(len(q)):
                    ores, ost = mo.find(q[i], k, approximate=mode == vc.MODE_MIH_APPROX, stop_mult=min(m, 4) if mode == vc.MODE_MIH_EXACT else 4)
                    g = got[i, : cnt[i]]
                    _check_contract(g, ores)
                    assert (st[i].radius, st[i].n_sub_reads, st[i].n_candidates) == (ost.radius, ost.n_sub_reads, ost.n_distinct), (lines, mode, i)
                    exp, _ = _canonical_mih(oracle, codes, q[i], m, k, ost.radius, False)
                    assert np.array_equal(g, exp)
            for t in range(m):                          # HashIndex -> Image_List views do not depend on the lines
                key = mo.key(codes[7], t)
                ids, _, total = e.get_bucket(t, key)
                assert np.array_equal(ids, mo.bucket(t, key))
    for mode in (vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
        assert np.array_equal(rows[("1", mode)], rows[("0", mode)])


@pytest.mark.parametrize("bits,m", [(128, 4), (64, 4)])
def test_launch_order_does_not_change_results(vc, oracle, monkeypatch, bits, m):
    """mih_order_kernel (batches of >= two residency waves of blocks: the query kernel's blocks take the queries in ascending
    order of their shell-0 bucket sizes, longest radius loops first): rows, counts and every statistic of a 4096-query batch are
    those of the same batch in batch order, and a sample of the queries equals the oracle's SearchWorker::find
    (search_worker.cc:159-218); with the directory lines (32-bit substrings) and with direct tables (16-bit)."""
    n, k, nq = 150_000, 20, 4096
    rng = np.random.default_rng(bits * 7 + m)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=500, max_flips=8)
    q = _near_queries(codes, nq, rng, 3)
    out = {}
    for order in ("1", "0"):
        monkeypatch.setenv("VC_MIH_ORDER", order)
        monkeypatch.setenv("VC_MIH_LINES", order)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            for mode in (vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
                got, cnt, st = e.search_knn(q, k, mode=mode, with_stats=True)
                out[(order, mode)] = (got.copy(), cnt.copy(), [(s.radius, s.n_sub_reads, s.n_local_reads, s.n_candidates, s.n_results) for s in st])
    for mode in (vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
        a, b = out[("1", mode)], out[("0", mode)]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    mo = oracle.MihOracle(codes, m, key_mode=1)
    got, cnt, st = out[("1", vc.MODE_MIH_EXACT)]
    for i in rng.choice(nq, size=24, replace=False):
        ores, ost = mo.find(q[i], k, stop_mult=4)
        _check_contract(got[i, : cnt[i]], ores)
        assert st[i][:2] == (ost.radius, ost.n_sub_reads) and st[i][3] == ost.n_distinct


@pytest.mark.parametrize("bits,m,nq", [(128, 4, 10_000), (128, 4, 20_000), (64, 4, 5_000)])
def test_launch_size_does_not_change_results(vc, oracle, monkeypatch, bits, m, nq):
    """A call's queries run in as few mih_query_kernel launches as VC_MIH_QTILE allows (default 16 384 per launch, equally filled:
    20 000 queries = 2 x 10 048 slots; the per-slot state grows with the batch): rows, counts and every statistic equal those of
    the same call in launches of 4 096, including the queries handed over to the multi-block shells and the far queries that the
    cost-model switch sends to the verify kernel; a sample equals the oracle's SearchWorker::find (search_worker.cc:159-218)."""
    n, k = 120_000, 20
    rng = np.random.default_rng(bits + nq)
    codes = oracle.gen_codes(n, bits, 35, kind=1, n_centres=400, max_flips=8)
    q = _near_queries(codes, nq, rng, 3)
    q[::997] = rng.integers(0, 256, size=q[::997].shape, dtype=np.uint8)      # a few far queries: hand-over / switch
    out = {}
    for tile in ("0", "4096"):
        monkeypatch.setenv("VC_MIH_QTILE", tile)
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(codes)
            e.build_index()
            for mode in (vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
                got, cnt, st = e.search_knn(q, k, mode=mode, with_stats=True)
                out[(tile, mode)] = (got.copy(), cnt.copy(), [(s.radius, s.n_sub_reads, s.n_local_reads, s.n_candidates, s.n_results) for s in st])
            small, scnt = e.search_knn(q[:100], k, mode=vc.MODE_MIH_EXACT)    # a small batch after a big one: the state is laid out anew
            assert np.array_equal(small, out[(tile, vc.MODE_MIH_EXACT)][0][:100]) and np.array_equal(scnt, out[(tile, vc.MODE_MIH_EXACT)][1][:100])
    for mode in (vc.MODE_MIH_EXACT, vc.MODE_MIH_APPROX):
        a, b = out[("0", mode)], out[("4096", mode)]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    mo = oracle.MihOracle(codes, m, key_mode=1)
    got, cnt, st = out[("0", vc.MODE_MIH_EXACT)]
    near = [i for i in rng.choice(nq, size=40, replace=False) if i % 997][:16]
    for i in near:
        ores, ost = mo.find(q[i], k, stop_mult=4)
        _check_contract(got[i, : cnt[i]], ores)
        assert st[i][:2] == (ost.radius, ost.n_sub_reads) and st[i][3] == ost.n_distinct
