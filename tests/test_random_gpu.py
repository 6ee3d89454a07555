"""Seeded random configurations: GPU (through the C ABI) against the oracle, linear and MIH, ragged sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SH = np.uint64(32)


def _cfg(i):
    rng = np.random.default_rng(1000 + i)
    bits = int(rng.choice([64, 128, 256]))
    s = int(rng.choice([8, 16, 32] if bits == 64 else ([16, 32] if bits == 128 else [32])))
    m = bits // s
    n = int(rng.integers(1, 50000))
    centres = int(rng.integers(1, 60))
    flips = int(rng.integers(0, 3 * m))
    nq = int(rng.integers(1, 40))
    # keep the k nearest inside the query's own cluster, or exact MIH must walk out to radius ~s/2 (2^s probes per
    # table on the CPU oracle); tiny databases are exercised with 8/16-bit substrings only
    k = int(rng.integers(1, 200))
    k = max(1, min(k, n // centres // 3))
    if n // centres < 6:
        s = min(s, 16) if bits == 64 else s
        m = bits // s
    return rng, bits, m, n, k, centres, flips, nq


@pytest.mark.parametrize("i", range(16))
def test_random_configuration(vc, oracle, i):
    rng, bits, m, n, k, centres, flips, nq = _cfg(i)
    id_base = int(rng.integers(0, 1000)) * 1000
    codes = oracle.gen_codes(n, bits, 50 + i, kind=1, n_centres=centres, max_flips=flips, first_id=id_base)
    q = codes[rng.integers(0, n, size=nq)].copy()
    for r in range(nq):                                         # 0..3 extra flips per query
        for b in rng.choice(bits, size=int(rng.integers(0, 4)), replace=False):
            q[r, b // 8] ^= np.uint8(1 << (b % 8))
    with vc.Engine(bits, capacity=n, n_tables=m, id_base=id_base, query_tile=int(rng.choice([1, 3, 8, 32]))) as e:
        if i % 2:
            e.add_codes(codes)
        else:
            e.add_synthetic(n, seed=50 + i, kind=1, n_centres=centres, max_flips=flips)
        lin, lcnt = e.search_knn(q, k)
        for r in range(nq):
            exp = oracle.linear_knn(codes, q[r], k, id_base=id_base)
            assert lcnt[r] == len(exp) and np.array_equal(lin[r, : lcnt[r]], exp), (i, r)
        e.build_index()
        mo = oracle.MihOracle(codes, m, key_mode=1, id_base=id_base)
        got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        for r in range(min(nq, 6)):                             # the CPU oracle enumerates shells one key at a time
            ores, ost = mo.find(q[r], k, stop_mult=min(m, 4))
            g = got[r, : cnt[r]]
            assert np.array_equal(g >> SH, np.sort(ores) >> SH), (i, r)
            assert (st[r].radius, st[r].n_sub_reads, st[r].n_candidates) == (ost.radius, ost.n_sub_reads, ost.n_distinct)
            if cnt[r]:
                dk = g[-1] >> SH
                assert set(g[(g >> SH) < dk].tolist()) == set(np.sort(ores)[(np.sort(ores) >> SH) < dk].tolist())
        rad = int(rng.integers(0, 2 * m + 3))
        a = e.search_radius(q[:5], rad, mode=vc.MODE_MIH_EXACT)
        b = e.search_radius(q[:5], rad, mode=vc.MODE_LINEAR)
        for r in range(min(nq, 5)):
            d = oracle.np_distances(codes, q[r])
            ids = np.nonzero(d <= rad)[0]
            exp = np.sort(oracle.pack(d[ids], ids.astype(np.uint64) + np.uint64(id_base)))
            assert np.array_equal(a[r], exp) and np.array_equal(b[r], exp), (i, r, rad)
