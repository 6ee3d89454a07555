"""GPU parity of the linear verify path (linear_search.cc:39-64) against the oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _queries(vo, codes, nq, rng, flips=6):
    """half near-duplicates of DB items (query-by-image use, image_search_client.h:23-25), half uniform."""
    nb = codes.shape[1]
    q = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8)
    for i in range(0, nq, 2):
        q[i] = codes[rng.integers(0, codes.shape[0])]
        for _ in range(rng.integers(0, flips + 1)):
            b = rng.integers(0, nb * 8)
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    return q


def _expect(vo, codes, q, k, id_base=0):
    out = np.full((q.shape[0], k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    cnt = np.zeros(q.shape[0], dtype=np.uint32)
    for i in range(q.shape[0]):
        r = vo.linear_knn(codes, q[i], k, id_base=id_base)
        out[i, : len(r)] = r
        cnt[i] = len(r)
    return out, cnt


def test_synthetic_generator_matches_oracle(vc, oracle):
    for bits, kind in ((64, 0), (128, 0), (128, 1), (256, 1), (512, 0)):
        n = 5000
        with vc.Engine(bits, capacity=n, id_base=1000) as e:
            e.add_synthetic(n, seed=34, kind=kind, n_centres=50, max_flips=9)
            ref = oracle.gen_codes(n, bits, 34, kind=kind, n_centres=50, max_flips=9, first_id=1000)
            for gid in (1000, 1001, 1063, 1064, 3333, 5999):
                assert np.array_equal(e.get_code(gid), ref[gid - 1000]), (bits, kind, gid)
            assert e.get_code(999) is None and e.get_code(6000) is None


@pytest.mark.parametrize("bits", [64, 128, 256, 512])
@pytest.mark.parametrize("n,k,kind", [(4096, 1, 0), (4097, 10, 0), (70001, 100, 0), (70001, 100, 1), (300000, 1000, 1)])
def test_linear_knn_bit_exact(vc, oracle, bits, n, k, kind):
    rng = np.random.default_rng(bits * 7 + n + k)
    codes = oracle.gen_codes(n, bits, 34, kind=kind, n_centres=200, max_flips=11)
    with vc.Engine(bits, capacity=n) as e:
        e.add_synthetic(n, seed=34, kind=kind, n_centres=200, max_flips=11)
        q = _queries(oracle, codes, 21, rng)   # 21: not a multiple of the query tile
        got, cnt = e.search_knn(q, k)
        exp, ecnt = _expect(oracle, codes, q, k)
        assert np.array_equal(cnt, ecnt)
        assert np.array_equal(got, exp)


def test_add_codes_rows_idbase_and_order(vc, oracle):
    rng = np.random.default_rng(5)
    n, bits, k = 33333, 128, 50
    codes = rng.integers(0, 256, size=(n, bits // 8), dtype=np.uint8)
    with vc.Engine(bits, capacity=n + 10, id_base=4_000_000_000) as e:
        e.add_codes(codes[:10000])
        e.add_codes(codes[10000:])
        assert len(e) == n
        q = _queries(oracle, codes, 5, rng)
        got, cnt = e.search_knn(q, k)
        exp, _ = _expect(oracle, codes, q, k, id_base=4_000_000_000)
        assert np.array_equal(got, exp)
        far, _ = e.search_knn(q, k, order=vc.ORDER_FARTHEST_FIRST)
        assert np.array_equal(far, exp[:, ::-1])
        # reference order (linear_search.cc:59-63) has the same distance sequence
        ref = oracle.linear_knn_ref(codes, q[0], k, id_base=4_000_000_000)
        assert np.array_equal(ref >> np.uint64(32), far[0] >> np.uint64(32))


def test_fewer_items_than_k(vc, oracle):
    rng = np.random.default_rng(9)
    codes = rng.integers(0, 256, size=(37, 16), dtype=np.uint8)
    with vc.Engine(128, capacity=64) as e:
        e.add_codes(codes)
        q = codes[:3].copy()
        got, cnt = e.search_knn(q, 100)
        exp, ecnt = _expect(oracle, codes, q, 100)
        assert list(cnt) == [37, 37, 37]
        assert np.array_equal(got, exp)


def test_many_exact_duplicates(vc, oracle):
    """ties at the k-th distance resolve to the smallest ids (canonical contract)."""
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, size=(50, 16), dtype=np.uint8)
    codes = np.repeat(base, 400, axis=0)           # 20000 items, 400 copies of each
    codes = codes[rng.permutation(codes.shape[0])]
    with vc.Engine(128, capacity=codes.shape[0]) as e:
        e.add_codes(codes)
        q = base[:4].copy()
        got, _ = e.search_knn(q, 100)
        exp, _ = _expect(oracle, codes, q, 100)
        assert np.array_equal(got, exp)


def test_second_bootstrap_stage(vc, oracle, monkeypatch):
    """stage 2 of the threshold bootstrap (normally only for >= 16 M codes) forced on a small database"""
    monkeypatch.setenv("VC_SAMPLE2", "150000")
    rng = np.random.default_rng(31)
    n, bits, k = 200001, 128, 100
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=50, max_flips=12)
    with vc.Engine(bits, capacity=n) as e:
        e.add_synthetic(n, seed=34, kind=1, n_centres=50, max_flips=12)
        q = _queries(oracle, codes, 10, rng)
        got, cnt = e.search_knn(q, k)
        exp, ecnt = _expect(oracle, codes, q, k)
        assert np.array_equal(cnt, ecnt) and np.array_equal(got, exp)


@pytest.mark.parametrize("device_recover", ["1", "0"])
def test_ring_overflow_recovery(vc, oracle, monkeypatch, device_recover):
    """more items at the k-th distance than the candidate ring holds: recovered on the device by a radix select over
    the position of the tied items (default), or -- VC_DEVICE_RECOVER=0, the fallback -- by repeating the scan with
    the packed bound of what did fit until nothing overflows; the answer is still the k smallest (dist, id)."""
    monkeypatch.setenv("VC_DEVICE_RECOVER", device_recover)
    rng = np.random.default_rng(12)
    base = rng.integers(0, 256, size=(10, 16), dtype=np.uint8)
    codes = np.repeat(base, 3000, axis=0)          # 30000 items, 3000 copies of each
    codes = codes[rng.permutation(codes.shape[0])]
    q = base[:3].copy()
    q[2, 5] ^= 0x10                                # 3000 ties at distance 1
    with vc.Engine(128, capacity=codes.shape[0], cand_cap=256) as e:   # ring = max(256, 4k) = 400 entries
        e.add_codes(codes)
        got, cnt = e.search_knn(q, 100)
        exp, ecnt = _expect(oracle, codes, q, 100)
        assert np.array_equal(cnt, ecnt) and np.array_equal(got, exp)


@pytest.mark.parametrize("device_recover", ["1", "0"])
def test_ring_overflow_with_tiny_ring_and_loose_early_entries(vc, oracle, monkeypatch, device_recover):
    """k = 1, ring of 4: the entries that fit arrive under the loose start threshold while thousands of duplicates
    (counted in the histogram, not stored) pull the final threshold below all of them -- the recovery bound must
    still come from what fitted (regression: the select pre-filter used to empty such a row)."""
    monkeypatch.setenv("VC_DEVICE_RECOVER", device_recover)
    n = 400_000
    codes = oracle.gen_codes(n, 256, 900, kind=1, n_centres=7, max_flips=0)    # 7 distinct codes, ~57 K copies each
    q = codes[[3, 1000, 77777, 399_999]].copy()
    q[1, 0] ^= 1
    with vc.Engine(256, capacity=n, query_tile=4, cand_cap=4) as e:
        e.add_codes(codes)
        for k in (1, 3):
            got, cnt = e.search_knn(q, k)
            exp, ecnt = _expect(oracle, codes, q, k)
            assert np.array_equal(cnt, ecnt) and np.array_equal(got, exp)


def _dev_search(e, q, k):
    import torch
    nq = q.shape[0]
    dq = torch.from_numpy(q).cuda()
    out = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    cnt = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    e.search_knn_dev(dq.data_ptr(), nq, k, out.data_ptr(), cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint64), cnt.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("cand_cap,k", [(4, 1), (4, 3), (64, 16), (256, 100), (1024, 256)])
def test_device_api_ring_overflow_is_exact(vc, oracle, cand_cap, k):
    """vc_search_knn_dev on a duplicate-heavy shard (7 distinct codes x 57 K copies) with rings of 4..1024 entries:
    every row overflows its ring and is recomputed exactly on the device, asynchronously (no UINT32_MAX marker left,
    counts == k), for 12 queries = three tiles, two recovery rounds (linear_search.cc:44-57 keeps the lowest ids)."""
    n = 400_000
    codes = oracle.gen_codes(n, 256, 900, kind=1, n_centres=7, max_flips=0)
    rng = np.random.default_rng(k)
    q = codes[rng.integers(0, n, size=12)].copy()
    q[1, 0] ^= 1                                   # ties at distance 1
    q[5, 7] ^= 0x81                                # ties at distance 2
    q[11] = rng.integers(0, 256, size=32, dtype=np.uint8)   # far from everything: ties at some large distance
    with vc.Engine(256, capacity=n, query_tile=4, cand_cap=cand_cap, id_base=1000) as e:
        e.add_codes(codes)
        got, cnt = _dev_search(e, q, k)
        exp, ecnt = _expect(oracle, codes, q, k, id_base=1000)
        assert np.array_equal(cnt, ecnt), cnt
        assert np.array_equal(got, exp)
        assert e.device_status() == 0
        host, hcnt = e.search_knn(q, k)            # the host API takes the same device path
        assert np.array_equal(host, exp) and np.array_equal(hcnt, ecnt)


def test_device_recovery_three_position_levels(vc, oracle):
    """5 M codes = 23 position bits = three radix levels (11 + 11 + 1 bits) of the tie select; mixed batch in which
    only some queries overflow"""
    n, bits = 5_000_000, 128
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, size=(3, bits // 8), dtype=np.uint8)
    codes = base[rng.integers(0, 3, size=n)]
    uni = oracle.gen_codes(1000, bits, 5)
    codes[rng.integers(0, n, size=1000)] = uni     # a sprinkle of distinct codes
    q = np.stack([base[0], base[2], uni[3], base[1]])
    q[3, 2] ^= 0x18
    for cand_cap, k in ((16, 5), (4096, 1000)):
        with vc.Engine(bits, capacity=n, cand_cap=cand_cap, query_tile=8) as e:
            e.add_codes(codes)
            got, cnt = _dev_search(e, q, k)
            exp, ecnt = _expect(oracle, codes, q, k)
            assert np.array_equal(cnt, ecnt) and np.array_equal(got, exp)
            assert e.device_status() == 0


def test_device_recovery_gives_up_cleanly_and_works_again(vc, oracle, monkeypatch):
    """A recovery grid that cannot meet (here: made to wait for a block that never comes, with a short spin bound) gives
    up: the rows keep count = UINT32_MAX, the sticky counter says so once -- and the engine is NOT poisoned: barrier
    words and step state are restored by the last block to leave, so the next call with an overflow recovers exactly
    (regression: the give-up path used to leave the barrier words dirty for good and reported nothing)."""
    monkeypatch.setenv("VC_RECOVER_TEST_FAIL", "2")         # the first two recover launches are sabotaged
    monkeypatch.setenv("VC_RECOVER_SPIN_LIMIT", "200")
    n = 400_000
    codes = oracle.gen_codes(n, 256, 900, kind=1, n_centres=7, max_flips=0)
    rng = np.random.default_rng(5)
    q = codes[rng.integers(0, n, size=4)].copy()
    q[1, 0] ^= 1
    k = 16
    exp, ecnt = _expect(oracle, codes, q, k)
    with vc.Engine(256, capacity=n, query_tile=4, cand_cap=64) as e:
        e.add_codes(codes)
        got, cnt = _dev_search(e, q, k)                     # launch 1: gives up
        assert np.all(cnt == np.uint32(0xFFFFFFFF))
        assert e.device_status() == 1 and e.device_status() == 0      # counted once, then cleared
        host, hcnt = e.search_knn(q, k)                     # launch 2 gives up as well: the host-driven fallback answers
        assert np.array_equal(host, exp) and np.array_equal(hcnt, ecnt)
        assert e.device_status() == 1
        for _ in range(2):                                  # launches 3, 4: the device recovery works again
            got, cnt = _dev_search(e, q, k)
            assert np.array_equal(cnt, ecnt) and np.array_equal(got, exp)
        assert e.device_status() == 0
        far = rng.integers(0, 256, size=(4, 32), dtype=np.uint8)      # and a step without overflow starts from clean state
        got, cnt = _dev_search(e, far, k)
        fexp, fcnt = _expect(oracle, codes, far, k)
        assert np.array_equal(got, fexp) and np.array_equal(cnt, fcnt)


@pytest.mark.parametrize("fold", ["1", "0"])
def test_ragged_groups_hand_back_clean_state(vc, oracle, monkeypatch, fold):
    """72 queries with tiles of 8 = one group of 64 + one of 8: the second group's bootstrap histograms sit 8 rows apart,
    not 64, and the step's last kernel must hand exactly those back zeroed -- the NEXT call's thresholds are cut from
    them (regression: residue of a ragged group made later thresholds too tight).  With the bootstrap cut folded into
    the verify prologue (default) and with the separate vc_tau_init_kernel launch (VC_TAU_FOLD=0)."""
    monkeypatch.setenv("VC_TAU_FOLD", fold)
    n, bits, k = 300_000, 128, 40
    rng = np.random.default_rng(72)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=500, max_flips=12)
    with vc.Engine(bits, capacity=n, query_tile=8) as e:
        e.add_codes(codes)
        for nq in (72, 72, 8, 130, 5):
            q = _queries(oracle, codes, nq, rng)
            got, cnt = e.search_knn(q, k)
            exp, ecnt = _expect(oracle, codes, q, k)
            assert np.array_equal(cnt, ecnt) and np.array_equal(got, exp), nq


def test_timing_reports_scan(vc):
    with vc.Engine(128, capacity=1 << 20) as e:
        e.add_synthetic(1 << 20, seed=1)
        q = np.zeros((4, 16), dtype=np.uint8)
        e.search_knn(q, 10)
        t = e.timing()
        assert t.scan_launches == 1 and t.calls == 1 and t.scan_bytes == (1 << 20) * 16 and t.scan_ms > 0


@pytest.mark.parametrize("bits,n,nq", [(64, 200_000, 150), (128, 50_000, 300), (256, 20_000, 40)])
def test_tile_and_tile_shape_follow_a_small_database(vc, oracle, bits, n, nq):
    """An engine whose query tile is left to it (vc_config.query_tile = 0) sizes the tile by the database -- a pass over a few MB
    is priced by its launches, so all of a call's queries (up to 512) share one pass -- and fits the verify kernel's lane tile to
    the few chunks there are (linear_tile, vc_scan_pick_shape).  Same rows as the fixed tile of 32 (five / ten / two passes
    here), as a tile of 8 (the headline's kernel form) and as the oracle's linear_search.cc:39-64 restatement."""
    rng = np.random.default_rng(bits + nq)
    codes = oracle.gen_codes(n, bits, 9, kind=1, n_centres=100, max_flips=9)
    q = _queries(oracle, codes, nq, rng)
    k = 64
    rows = {}
    for tile in (0, 32, 8):
        with vc.Engine(bits, capacity=n, query_tile=tile) as e:
            e.add_codes(codes)
            rows[tile] = e.search_knn(q, k)
    for tile in (32, 8):
        assert np.array_equal(rows[0][0], rows[tile][0]) and np.array_equal(rows[0][1], rows[tile][1])
    exp, ecnt = _expect(oracle, codes, q[:24], k)
    assert np.array_equal(rows[0][0][:24], exp) and np.array_equal(rows[0][1][:24], ecnt)
