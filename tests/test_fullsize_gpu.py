"""BASELINE.json shapes at FULL size on one MI355X.  Checked against the CPU oracle itself -- the database is
regenerated on the host slab by slab from the shared generator definition and scanned by the oracle's
linear_search.cc restatement on a worker pool (exact packed rows for a subset of the queries) -- and through
size-independent properties for the rest: planted neighbours are found at their exact distance, every reported
distance is recomputed from the stored code, results are ascending and reproducible, independent GPU paths
agree (MIH == linear scan; one engine == two half-database engines + merge; query tile 1 == query tile 8)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SH = np.uint64(32)
MASK = np.uint64(0xFFFFFFFF)


def _flip(code, bits, rng):
    c = code.copy()
    for b in bits:
        c[b // 8] ^= np.uint8(1 << (b % 8))
    return c


def _check_rows(e, q, rows, counts):
    for i in range(q.shape[0]):
        r = rows[i, : counts[i]]
        assert np.all(r[1:] > r[:-1])                       # ascending, no duplicates
        for j in (0, len(r) // 2, len(r) - 1):              # distances are what the stored code says
            code = e.get_code(int(r[j] & MASK))
            assert int(np.unpackbits(code ^ q[i]).sum()) == int(r[j] >> SH)


def test_config1_plumbing_64bit_1M(vc, oracle):
    """BASELINE configs[0] at its stated shape: linear_search.cc:39-64 over 64-bit codes, N = 2^20,
    the reference's 200-query cap (distributed_image_search.cc:83-84), k = 100.  Every row of the HIP scan equals the
    oracle's canonical rows; the farthest-first order (what linear_search.cc:59-63 prints) carries the distance sequence
    of the reference-order restatement; ids below the k-th distance are the reference heap's."""
    n, bits, nq, k = 1 << 20, 64, 200, 100
    rng = np.random.default_rng(1)
    codes = oracle.gen_codes(n, bits, 34)
    q = rng.integers(0, 256, size=(nq, bits // 8), dtype=np.uint8)
    for i in range(0, nq, 4):                                # every fourth query is a near-duplicate of a stored image
        q[i] = _flip(codes[int(rng.integers(0, n))], rng.choice(bits, size=int(rng.integers(0, 9)), replace=False), rng)
    with vc.Engine(bits, capacity=n) as e:
        e.add_synthetic(n, seed=34)
        assert np.array_equal(e.get_code(n - 1), codes[n - 1])
        rows, cnt = e.search_knn(q, k)
        far, fcnt = e.search_knn(q, k, order=vc.ORDER_FARTHEST_FIRST)
        with vc.Engine(bits, capacity=n) as h:               # the same records ingested from host memory (vc_add_codes)
            h.add_codes(codes)
            rows_h, _ = h.search_knn(q, k)
    assert np.all(cnt == k) and np.all(fcnt == k)
    assert np.array_equal(rows, rows_h)
    with oracle.Pool() as pool:
        exp, ecnt = pool.linear_knn(codes, q, k)
    assert np.all(ecnt == k) and np.array_equal(rows, exp)
    assert np.array_equal(far, rows[:, ::-1])
    for i in range(0, nq, 8):                                # the reference-order restatement (std::priority_queue, strict >)
        ref = oracle.linear_knn_ref(codes, q[i], k)
        assert np.array_equal(far[i] >> SH, ref >> SH)       # same distances, farthest first
        dk = rows[i, -1] >> SH
        assert set(ref[(ref >> SH) < dk].tolist()) == set(rows[i][(rows[i] >> SH) < dk].tolist())


def test_config3_top100_over_1e9_codes_128bit(vc, oracle):
    n, bits, k = 1_000_000_000, 128, 100
    rng = np.random.default_rng(3)
    with vc.Engine(bits, capacity=n, query_tile=8) as e:
        e.add_synthetic(n, seed=34)
        plant = [int(x) for x in rng.integers(0, n, size=6)]
        nflip = [0, 1, 3, 8, 13, 21]
        q = np.stack([_flip(e.get_code(g), rng.choice(bits, size=f, replace=False), rng) for g, f in zip(plant, nflip)]
                     + [rng.integers(0, 256, size=bits // 8, dtype=np.uint8) for _ in range(2)])   # + two with no planted neighbour
        rows, cnt = e.search_knn(q, k)
        assert np.all(cnt == k)
        for i, (g, f) in enumerate(zip(plant, nflip)):      # the planted item is the nearest neighbour, at f bits
            assert int(rows[i, 0] & MASK) == g and int(rows[i, 0] >> SH) == f
            assert int(rows[i, 1] >> SH) > 21               # uniform 128-bit codes: everything else is far away
        _check_rows(e, q, rows, cnt)
        # the oracle over the SAME 1e9 codes (linear_search.cc:39-64 restated; the database regenerated on the host slab
        # by slab): exact packed rows of two planted and the two uniform queries -- a true neighbour dropped anywhere in
        # the 16 GB would show here
        sel = [1, 4, 6, 7]
        with oracle.Pool() as pool:
            exp = oracle.linear_knn_slabbed(pool, n, bits, 34, q[sel], k)
        assert np.array_equal(rows[sel], exp)
        again, _ = e.search_knn(q, k)                       # reproducible despite racing threshold updates
        assert np.array_equal(rows, again)
        one_by_one = np.stack([e.search_knn(q[i:i + 1], k)[0][0] for i in range(2)])   # tile of 1 == tile of 8
        assert np.array_equal(one_by_one, rows[:2])
        t = e.timing()
        assert t.scan_bytes == t.scan_launches * n * 16
    # top-k of the union == merge of the halves' top-k (the multi-GPU identity), on real engines
    import torch
    halves = []
    for h in range(2):
        eh = vc.Engine(bits, capacity=n // 2, id_base=h * (n // 2), query_tile=8)
        eh.add_synthetic(n // 2, seed=34)
        halves.append(eh)
    dq = torch.from_numpy(q).cuda()
    gathered = torch.empty((2, 8, k), dtype=torch.int64, device="cuda")
    cnts = torch.empty((8,), dtype=torch.int32, device="cuda")
    for h, eh in enumerate(halves):
        eh.search_knn_dev(dq.data_ptr(), 8, k, gathered[h].data_ptr(), cnts.data_ptr())
        eh.timing()                                          # synchronises with the engine's stream
    out = torch.empty((8, k), dtype=torch.int64, device="cuda")
    vc.merge_topk_dev(gathered.data_ptr(), 2, 8, k, out.data_ptr(), cnts.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), rows)
    for eh in halves:
        eh.close()


def test_config2_radius8_mih_over_1e8_codes_64bit(vc, oracle):
    n, bits, m, radius = 100_000_000, 64, 2, 8
    rng = np.random.default_rng(2)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_synthetic(n, seed=34)
        e.build_index()
        plant = [int(x) for x in rng.integers(0, n, size=16)]
        nflip = [int(x) for x in rng.integers(0, radius + 1, size=16)]
        q = np.stack([_flip(e.get_code(g), rng.choice(bits, size=f, replace=False), rng) for g, f in zip(plant, nflip)]
                     + [rng.integers(0, 256, size=bits // 8, dtype=np.uint8)])      # + one with (almost surely) no neighbour
        mih = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT)
        lin = e.search_radius(q, radius, mode=vc.MODE_LINEAR)
        for i, (g, f) in enumerate(zip(plant, nflip)):
            assert np.array_equal(mih[i], lin[i])            # hash-probe path == full scan
            assert (np.uint64(f) << SH) | np.uint64(g) in mih[i]
            assert np.all((mih[i] >> SH) <= radius) and np.all(mih[i][1:] > mih[i][:-1])
        assert np.array_equal(mih[16], lin[16])
        # the oracle's brute force over the SAME 1e8 codes (compute_hamming_dist of every record, image_tools.h:21-33):
        # "all within 8" of six of the queries equals both the MIH and the scan result
        sel = [0, 3, 7, 11, 15, 16]
        with oracle.Pool() as pool:
            exp = oracle.linear_radius_slabbed(pool, n, bits, 34, q[sel], radius)
        for j, i in enumerate(sel):
            assert np.array_equal(mih[i], exp[j]) and np.array_equal(lin[i], exp[j])
        # exact k-NN through MIH agrees with the scan on the distances (ties at the k-th distance may differ)
        got, cnt, st = e.search_knn(q[:4], 3, mode=vc.MODE_MIH_EXACT, with_stats=True)
        ref, _ = e.search_knn(q[:4], 3, mode=vc.MODE_LINEAR)
        assert np.array_equal(got >> SH, ref >> SH)
        # key -> bucket -> ids round trip for the planted items (rule a12)
        for g in plant[:4]:
            code = e.get_code(g)
            for t in range(m):
                key = int.from_bytes(bytes(code[t * 4:(t + 1) * 4]), "little")
                ids, codes, total = e.get_bucket(t, key)
                assert g in ids and np.array_equal(codes[list(ids).index(g)], code)


def test_config5_shape_256bit_4096_query_tile(vc):
    """LDS query-tile stress (4096 x 32 B = 128 KiB of queries + 16 KiB of thresholds in LDS, 512-thread blocks):
    one pass over the database for all 4096 queries must equal 256 passes of 16."""
    n, bits, k, nq = 3_000_000, 256, 100, 4096
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, size=(nq, bits // 8), dtype=np.uint8)
    with vc.Engine(bits, capacity=n, query_tile=4096) as big, vc.Engine(bits, capacity=n, query_tile=16) as small:
        for e in (big, small):
            e.add_synthetic(n, seed=34)
        a, ca = big.search_knn(q, k)
        tb = big.timing()
        b, cb = small.search_knn(q, k)
        ts = small.timing()
        assert tb.scan_launches == 1 and ts.scan_launches == 256
        assert np.array_equal(a, b) and np.array_equal(ca, cb)
        _check_rows(big, q[:3], a[:3], ca[:3])


def test_config5_tile_4096_against_the_oracle(vc, oracle):
    """BASELINE configs[4]'s query tile (4096 x 256-bit queries in LDS, 512-thread blocks, one pass over the database)
    against the CPU oracle itself -- not against another HIP tile: 512 of the 4096 rows, uniform and near-duplicate
    queries (linear_search.cc:44-57)."""
    n, bits, k, nq = 250_000, 256, 100, 4096
    rng = np.random.default_rng(55)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=900, max_flips=20)
    q = rng.integers(0, 256, size=(nq, bits // 8), dtype=np.uint8)
    near = rng.choice(nq, size=1024, replace=False)
    q[near] = codes[rng.integers(0, n, size=1024)]
    for i in near[::2]:
        q[i] = _flip(q[i], rng.choice(bits, size=int(rng.integers(1, 30)), replace=False), rng)
    with vc.Engine(bits, capacity=n, query_tile=4096) as e:
        e.add_synthetic(n, seed=34, kind=1, n_centres=900, max_flips=20)
        rows, cnt = e.search_knn(q, k)
        t = e.timing()
        assert t.scan_launches == 1 and t.scan_bytes == n * 32      # ONE pass for all 4096 queries
        assert np.all(cnt == k)
        check = np.concatenate([[0, 1, nq - 1], near[:253], rng.choice(nq, size=256, replace=False)])
        for i in check:
            assert np.array_equal(rows[i], oracle.linear_knn(codes, q[i], k)), i


def test_config5_per_gpu_share_5e8_codes_4096_queries(vc):
    """configs[4] at the size one of its 8 GPUs holds (5e8 x 256-bit codes = 16 GB, 4096 queries per pass), through
    size-independent properties: planted neighbours come back first at their exact distance, rows ascend, reported
    distances are what the stored codes say, and the 4096-query tile agrees with the 8-query tile of a second engine."""
    n, bits, k, nq = 500_000_000, 256, 100, 4096
    rng = np.random.default_rng(6)
    q = rng.integers(0, 256, size=(nq, bits // 8), dtype=np.uint8)
    with vc.Engine(bits, capacity=n, query_tile=4096) as e:
        e.add_synthetic(n, seed=34)
        slots = [0, 1, 511, 512, 2047, 4095, 77, 3000]
        plant = [int(x) for x in rng.integers(0, n, size=len(slots))]
        nflip = [0, 1, 2, 5, 9, 17, 33, 60]
        for s, g, f in zip(slots, plant, nflip):
            q[s] = _flip(e.get_code(g), rng.choice(bits, size=f, replace=False), rng)
        rows, cnt = e.search_knn(q, k)
        t = e.timing()
        assert t.scan_launches == 1 and t.scan_bytes == n * 32
        assert np.all(cnt == k)
        assert np.all(rows[:, 1:] > rows[:, :-1])
        for s, g, f in zip(slots, plant, nflip):
            assert int(rows[s, 0] & MASK) == g and int(rows[s, 0] >> SH) == f
        _check_rows(e, q[slots[:4]], rows[slots[:4]], cnt[slots[:4]])
        _check_rows(e, q[[5, 2500]], rows[[5, 2500]], cnt[[5, 2500]])
    sub = slots + [5, 2500, 4000]
    with vc.Engine(bits, capacity=n, query_tile=8) as e8:
        e8.add_synthetic(n, seed=34)
        r8, c8 = e8.search_knn(q[sub], k)
        assert np.array_equal(r8, rows[sub]) and np.all(c8 == k)


def test_mih_exact_top100_over_1e9_clustered_codes_128bit(vc, oracle):
    """The metric's size through the north star's engine: SearchWorker::find (search_worker.cc:159-218) exact top-100 by
    multi-index hashing over 1e9 clustered 128-bit codes, m = 4 tables of 32-bit substrings (34 GB of index next to the
    16 GB of codes; the {id, code} record copies do not fit a third of the memory here, so entries are gathered through
    ids[] -> code columns).  Distances against the oracle's linear scan of the same 1e9 codes for four queries, against
    the HIP scan for all, ids below the k-th distance identical, statistics equal to the replayed stop rule."""
    n, bits, k, m = 1_000_000_000, 128, 100, 4
    nc, mf = n // 1000, 11
    rng = np.random.default_rng(9)
    nq = 64
    with vc.Engine(bits, capacity=n, n_tables=m, query_tile=8) as e:
        e.add_synthetic(n, seed=34, kind=vc.SYNTH_CLUSTERED, n_centres=nc, max_flips=mf)
        e.build_index()
        plant = [int(x) for x in rng.integers(0, n, size=nq)]
        nflip = [int(x) for x in rng.integers(0, 5, size=nq)]
        q = np.stack([_flip(e.get_code(g), rng.choice(bits, size=f, replace=False), rng) for g, f in zip(plant, nflip)])
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        lin, lcnt = e.search_knn(q, k, mode=vc.MODE_LINEAR)
        assert np.all(cnt == k) and np.all(lcnt == k)
        shell = [1, 32, 496, 4960, 35960, 201376, 906192, 3365856, 10518300]        # C(32, r)
        for i in range(nq):
            assert np.array_equal(got[i] >> SH, lin[i] >> SH)                       # same distance multiset as the scan
            D = int(got[i, -1] >> SH)
            below = got[i][(got[i] >> SH) < np.uint64(D)]
            assert np.array_equal(below, lin[i][: len(below)])                      # identical below the k-th distance
            pk = (np.uint64(nflip[i]) << SH) | np.uint64(plant[i])
            assert pk in got[i] or pk > got[i, -1]                                  # the planted item, unless k nearer ones exist
            # stop rule replayed (search_worker.cc:201-205): shell floor(D/4), or one earlier when D is a multiple of 4
            r0 = D // 4
            assert st[i].radius in ((r0 - 1, r0) if D % 4 == 0 and D else (r0,))
            assert st[i].n_sub_reads == sum(shell[: st[i].radius + 1]) and st[i].n_local_reads == 0
            assert st[i].n_candidates >= k
        _check_rows(e, q[:6], got[:6], cnt[:6])
        sel = [0, 1, 2, 3]
        with oracle.Pool() as pool:
            exp = oracle.linear_knn_slabbed(pool, n, bits, 34, q[sel], k, kind=1, n_centres=nc, max_flips=mf)
        assert np.array_equal(lin[sel], exp)                                        # the HIP scan IS the oracle's row
        assert np.array_equal(got[sel] >> SH, exp >> SH)


def test_config2_m4_variant_streams_its_buckets_over_1e8_codes(vc, oracle):
    """BASELINE configs[1] with the reference's default table count: 64-bit codes in m = 4 substrings of 16 bit, buckets of
    ~1 526 entries each -- the radius-8 search streams 188 buckets per query from the bucket-order code copies
    (mih_bucket_stream_kernel).  MIH == HIP scan for all queries, == the oracle's brute force over the same 1e8 codes."""
    n, bits, m, radius = 100_000_000, 64, 4, 8
    rng = np.random.default_rng(24)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_synthetic(n, seed=34)
        e.build_index()
        plant = [int(x) for x in rng.integers(0, n, size=24)]
        nflip = [int(x) for x in rng.integers(0, radius + 1, size=24)]
        q = np.stack([_flip(e.get_code(g), rng.choice(bits, size=f, replace=False), rng) for g, f in zip(plant, nflip)])
        mih = e.search_radius(q, radius, mode=vc.MODE_MIH_EXACT)
        t = e.timing()
        assert t.mih_launches >= 1 and t.mih_entries > 24 * 150_000       # ~287 K bucket entries per query were streamed
        lin = e.search_radius(q, radius, mode=vc.MODE_LINEAR)
        for i, (g, f) in enumerate(zip(plant, nflip)):
            assert np.array_equal(mih[i], lin[i])
            assert (np.uint64(f) << SH) | np.uint64(g) in mih[i]
        sel = [0, 5, 11, 23]
        with oracle.Pool() as pool:
            exp = oracle.linear_radius_slabbed(pool, n, bits, 34, q[sel], radius)
        for j, i in enumerate(sel):
            assert np.array_equal(mih[i], exp[j])


def test_config4_1e9_codes_sharded_8_ways_through_the_c_abi(vc, oracle):
    """BASELINE configs[3] in its shard arithmetic -- 128-bit codes, 1e9-code database split by id range into 8 shards of
    125 M, per-shard top-100, one exchange, device merge -- through vc_sharded_* with the eight shards on the ONE GPU of
    the test box (peer-copy exchange in place of the all-gather over xGMI).  Rows == the oracle's scan of the whole 1e9
    codes for two queries, == a single engine holding everything for all."""
    n, bits, k = 1_000_000_000, 128, 100
    rng = np.random.default_rng(31)
    q = rng.integers(0, 256, size=(8, bits // 8), dtype=np.uint8)
    with vc.ShardedEngine(bits, capacity=n, n_shards=8, devices=[0], query_tile=8) as s:
        s.add_synthetic(n, seed=34)
        assert len(s) == n and s.exchange == vc.EXCHANGE_PEER_COPY
        assert [s.shard_range(g) for g in range(8)] == [(g * 125_000_000, 125_000_000) for g in range(8)]
        plant = [int(x) for x in rng.integers(0, n, size=4)]            # near-duplicates of items of four different shards
        for i, g in enumerate(plant):
            q[i] = _flip(s.get_code(g), rng.choice(bits, size=i + 1, replace=False), rng)
        rows, cnt = s.search_knn(q, k)
        assert np.all(cnt == k)
        for i, g in enumerate(plant):
            assert int(rows[i, 0]) == ((i + 1) << 32 | g)
    with vc.Engine(bits, capacity=n, query_tile=8) as e:
        e.add_synthetic(n, seed=34)
        one, _ = e.search_knn(q, k)
    assert np.array_equal(rows, one)
    with oracle.Pool() as pool:
        exp = oracle.linear_knn_slabbed(pool, n, bits, 34, q[[0, 7]], k)
    assert np.array_equal(rows[[0, 7]], exp)


def test_config5_4e9_codes_256bit_4096_queries_sharded_8_ways_on_one_device(vc, oracle):
    """BASELINE configs[4] at FULL size: 256-bit codes, 4e9-code database (128 GB), 4096 queries per batch in one LDS query
    tile per shard, eight id-range shards of 5e8 codes -- all eight resident on the one 288 GB GPU of the test box, searched
    through vc_sharded_* (what eight GPUs do side by side, one after the other here; peer-copy exchange, device merge).
    Planted neighbours from different shards come back first at their exact distance with their 32-bit global ids (up to
    4e9 - 1), rows ascend, and two rows equal the oracle's scan of all 4e9 codes."""
    n, bits, k, nq = 4_000_000_000, 256, 100, 4096
    rng = np.random.default_rng(44)
    q = rng.integers(0, 256, size=(nq, bits // 8), dtype=np.uint8)
    with vc.ShardedEngine(bits, capacity=n, n_shards=8, devices=[0], query_tile=4096) as s:
        s.add_synthetic(n, seed=34)
        slots = [0, 1, 777, 2048, 4095]
        plant = [5, 499_999_999, 500_000_000, 3_141_592_653, n - 1]   # shard edges, an id beyond 2^31, the last id
        nflip = [0, 3, 7, 12, 30]
        for sl, g, f in zip(slots, plant, nflip):
            q[sl] = _flip(s.get_code(g), rng.choice(bits, size=f, replace=False), rng)
        rows, cnt = s.search_knn(q, k)
        assert np.all(cnt == k) and np.all(rows[:, 1:] > rows[:, :-1])
        for sl, g, f in zip(slots, plant, nflip):
            assert int(rows[sl, 0] & MASK) == g and int(rows[sl, 0] >> SH) == f
        for sl in (5, 3000):                                          # reported distances are what the stored codes say
            for j in (0, k - 1):
                code = s.get_code(int(rows[sl, j] & MASK))
                assert int(np.unpackbits(code ^ q[sl]).sum()) == int(rows[sl, j] >> SH)
    sel = [777, 3000]
    with oracle.Pool() as pool:
        exp = oracle.linear_knn_slabbed(pool, n, bits, 34, q[sel], k)
    assert np.array_equal(rows[sel], exp)
