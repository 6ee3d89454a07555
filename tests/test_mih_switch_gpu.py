"""Cost-model switch of the exact MIH k-NN loop: queries whose remaining shells cost more probes than a
streaming pass are answered by the verify kernel, and the stop rule of search_worker.cc:201-205 is REPLAYED on the scan's
candidates -- rows, radius, n_sub_reads and the distinct-candidate count must be exactly what the radius loop
(oracle: MihOracle.find) produces, including the shell-early stop when the k-th distance is a multiple of the table
count."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SH = np.uint64(32)


def _near(codes, rng, nq, flips):
    q = codes[rng.integers(0, codes.shape[0], size=nq)].copy()
    for i in range(nq):
        for b in rng.choice(codes.shape[1] * 8, size=int(rng.integers(0, flips + 1)), replace=False):
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    return q


def _canonical(vo, codes, q, m, k, radius):
    seen = vo.np_sub_distances(codes, q, m).min(axis=1) <= radius
    d = vo.np_distances(codes, q)
    ids = np.arange(codes.shape[0], dtype=np.uint64)
    return np.sort(vo.pack(d[seen], ids[seen]))[:k], int(seen.sum())


@pytest.mark.parametrize("bits,m,k", [(128, 4, 100), (128, 4, 7), (64, 2, 20), (64, 4, 16), (256, 8, 50), (64, 8, 5)])
def test_forced_switch_reproduces_the_radius_loop(vc, oracle, monkeypatch, bits, m, k):
    """VC_MIH_HOST_LOOP=1 + VC_MIH_SWITCH=2: every query goes through scan + replay from shell 0 on"""
    monkeypatch.setenv("VC_MIH_HOST_LOOP", "1")
    monkeypatch.setenv("VC_MIH_SWITCH", "2")
    n = 40000
    rng = np.random.default_rng(bits * 7 + m + k)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=60, max_flips=2 * m)
    # far queries: uniform random ones where the ORACLE's shell enumeration stays cheap (<= 16-bit substrings: at most 2^16
    # keys per table), else database items with up to 24 flips (k-th distance up to ~25: shells 0..6 of 32-bit substrings)
    far = (rng.integers(0, 256, size=(4, bits // 8), dtype=np.uint8) if bits // m <= 16 else _near(codes, rng, 4, min(6 * m, 24)))
    q = np.concatenate([_near(codes, rng, 20, m + 2), far])
    mo = oracle.MihOracle(codes, m, key_mode=1)
    s = bits // m
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        t = e.timing()
        assert t.scan_launches >= 1                                   # the verify kernel really answered them
        early = 0
        for i in range(len(q)):
            ores, ost = mo.find(q[i], k, stop_mult=min(m, 4))
            o = np.sort(ores)
            assert cnt[i] == k and np.array_equal(got[i] >> SH, o >> SH)
            assert (st[i].radius, st[i].n_sub_reads, st[i].n_local_reads) == (ost.radius, ost.n_sub_reads, 0), i
            assert st[i].n_sub_reads == sum(math.comb(s, r) for r in range(ost.radius + 1))
            exp, n_seen = _canonical(oracle, codes, q[i], m, k, ost.radius)
            assert np.array_equal(got[i], exp), i
            assert st[i].n_candidates == ost.n_distinct == n_seen
            D = int(o[-1] >> SH)
            early += m <= 4 and D > 0 and D % m == 0 and ost.radius == D // m - 1
        if (bits, m, k) == (128, 4, 100):
            assert early > 0                                          # the one-shell-early stop occurs in this data set


def test_dev_api_and_fewer_items_than_k(vc, oracle, monkeypatch):
    """device-pointer API (no statistics pass), k larger than the database (the loop runs to its last shell)"""
    import torch
    monkeypatch.setenv("VC_MIH_HOST_LOOP", "1")
    monkeypatch.setenv("VC_MIH_SWITCH", "2")
    n, bits, m, k = 30, 128, 4, 40
    codes = oracle.gen_codes(n, bits, 3)
    q = codes[:3].copy()
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        for i in range(3):
            exp = oracle.linear_knn(codes, q[i], k)
            assert cnt[i] == n and np.array_equal(got[i, :n], exp) and st[i].radius == 32 and st[i].n_candidates == n
        dq = torch.from_numpy(q).cuda()
        out = torch.zeros((3, k), dtype=torch.int64, device="cuda")
        c = torch.zeros((3,), dtype=torch.int32, device="cuda")
        e.search_knn_dev(dq.data_ptr(), 3, k, out.data_ptr(), c.data_ptr(), mode=vc.MODE_MIH_EXACT)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), got)


def test_cost_model_switches_uniform_queries_and_leaves_near_duplicates(vc, oracle):
    """no knobs: on 3 M uniform 128-bit codes a uniform-random query needs shells up to r ~ 9 (1e8 probes per table) -- the
    cost model hands it to the verify kernel after the in-kernel shells; near-duplicate queries of a clustered database
    never get there.  Results and statistics equal the canonical rule at the reported radius, the radius equals the
    replayed stop rule, n_sub_reads the closed form."""
    n, bits, m, k = 3_000_000, 128, 4, 10
    rng = np.random.default_rng(11)
    codes = oracle.gen_codes(n, bits, 34)
    q = rng.integers(0, 256, size=(6, bits // 8), dtype=np.uint8)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_synthetic(n, seed=34)
        e.build_index()
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        t = e.timing()
        assert t.scan_launches >= 1
        lin, _ = e.search_knn(q, k)
        for i in range(len(q)):
            D = int(lin[i, -1] >> SH)
            assert st[i].radius in ((D // 4 - 1, D // 4) if D % 4 == 0 else (D // 4,))
            exp, n_seen = _canonical(oracle, codes, q[i], m, k, st[i].radius)
            assert np.array_equal(got[i], exp) and st[i].n_candidates == n_seen
            assert st[i].n_sub_reads == sum(math.comb(32, r) for r in range(st[i].radius + 1))
            assert np.array_equal(got[i] >> SH, lin[i] >> SH)
    cl = oracle.gen_codes(200_000, bits, 34, kind=1, n_centres=200, max_flips=8)
    with vc.Engine(bits, capacity=len(cl), n_tables=m) as e:
        e.add_codes(cl)
        e.build_index()
        e.search_knn(_near(cl, rng, 64, 3), k, mode=vc.MODE_MIH_EXACT)
        assert e.timing().scan_launches == 0                          # cheap queries stay in the radius loop
