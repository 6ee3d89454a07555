"""Test infrastructure (GPU box, run by hand; not collected by pytest): seeded parity campaign of the MIH path
(probe, bitmap, verify, commit, stop rule, radius search) against the oracle's restatement of search_worker.cc, beyond the regular suite: every substring width, the reference
quirk flags, exact and approximate mode, attached bitmap, ragged sizes, id bases, big k, duplicate-heavy data.
Checks per query: the SURVEY 8c contract (distance multiset + id set below the k-th distance), radius / n_sub_reads /
n_local_reads / distinct-candidate statistics, and the engine's canonical rule exactly.
usage: python tests/campaign/parity_campaign_mih.py [n_cases=200] [seed0=0]   (test infrastructure: imports oracle/)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vc_oracle as oracle  # noqa: E402
from verticut_amd import engine as vc  # noqa: E402

SH = np.uint64(32)


def reach(codes, q, m, signext):
    sub = oracle.np_sub_distances(codes, q, m).astype(np.int64)
    if signext:
        nlb = codes.shape[1] // m
        x = np.bitwise_xor(codes, q[None, :]).reshape(codes.shape[0], m, nlb)
        sub[(x[:, :, nlb - 1] & 0x80) != 0] = 10 ** 6
    return sub.min(axis=1)


def canonical(codes, q, m, k, radius, signext, id_base):
    seen = reach(codes, q, m, signext) <= radius
    d = oracle.np_distances(codes, q)
    ids = np.arange(codes.shape[0], dtype=np.uint64) + np.uint64(id_base)
    return np.sort(oracle.pack(d[seen], ids[seen]))[:k]


def case(i):
    rng = np.random.default_rng(9000 + i)
    bits = int(rng.choice([64, 64, 128, 128, 256]))
    s = int(rng.choice([8, 16, 32] if bits == 64 else ([16, 32] if bits == 128 else [32])))
    m = bits // s
    n = int(rng.integers(1, 1 << int(rng.integers(8, 21))))
    centres = int(rng.integers(1, max(2, n // 50)))
    centres = min(centres, 4000)
    flips = int(rng.integers(0, 2 * m + 2))
    nq = int(rng.integers(1, 12))
    k = int(rng.choice([1, 5, 20, 100, 300]))
    k = max(1, min(k, n // centres // 2 if n // centres >= 2 else 1))   # keep the k nearest inside a cluster (see tests)
    flag = str(rng.choice(["", "", "signext", "literal4", "bitmap"]))
    if s == 32 and flag == "signext":
        flag = ""                      # the sign-extension quirk only exists below 32 bits
    approx = bool(rng.integers(0, 4) == 0) and flag in ("", "bitmap")
    # the CPU oracle walks shells one key at a time: keep every case inside a few shells.  Approximate mode stops at
    # 20k seen candidates (search_worker.h:14), so the query's cluster must hold that many; 32-bit substrings get few flips.
    if approx and 40 * k > n // centres:
        approx = False
    if s == 32:
        flips = min(flips, 5)
    id_base = int(rng.integers(0, 1 << 16))
    return rng, bits, s, m, n, centres, flips, nq, k, flag, approx, id_base


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build(ref=False)
    t0 = time.time()
    checked = 0
    for i in range(seed0, seed0 + n_cases):
        rng, bits, s, m, n, centres, flips, nq, k, flag, approx, id_base = case(i)
        flags = {"": 0, "signext": vc.FLAG_REF_SIGNEXT_KEYS, "literal4": vc.FLAG_REF_STOP_LITERAL4,
                 "bitmap": vc.FLAG_USE_BITMAP}[flag]
        signext = flag == "signext"
        stop_mult = 4 if (flag == "literal4" or approx) else min(m, 4)
        codes = oracle.gen_codes(n, bits, 300 + i, kind=1, n_centres=centres, max_flips=flips, first_id=id_base)
        q = codes[rng.integers(0, n, size=nq)].copy()
        for r in range(nq):
            for b in rng.choice(bits, size=int(rng.integers(0, 3)), replace=False):
                q[r, b // 8] ^= np.uint8(1 << (b % 8))
        mo = oracle.MihOracle(codes, m, key_mode=0 if signext else 1, id_base=id_base)
        desc = "bits=%3d s=%2d n=%7d centres=%4d flips=%2d k=%3d nq=%2d flag=%-8s approx=%d" % (
            bits, s, n, centres, flips, k, nq, flag, approx)
        # every fifth case sends the exact loop through the verify kernel + replayed stop rule (the cost-model switch, forced),
        # another fifth sends <= 16-bit radius searches through the bucket streaming kernel (knobs are read at vc_create)
        for kn in ("VC_MIH_SWITCH", "VC_MIH_HOST_LOOP", "VC_MIH_STREAM", "VC_MIH_LINES"):
            os.environ.pop(kn, None)
        if s == 32 and i % 3 == 0:       # directory lines (built by default only from 3e8 records on): every third 32-bit case
            os.environ["VC_MIH_LINES"] = "1"
            desc += " [lines]"
        if i % 5 == 3:
            os.environ.update(VC_MIH_SWITCH="2", VC_MIH_HOST_LOOP="1")
            desc += " [switch]"
        elif i % 5 == 4:
            os.environ["VC_MIH_STREAM"] = "2"
            desc += " [stream]"
        with vc.Engine(bits, capacity=n, n_tables=m, flags=flags, id_base=id_base,
                       query_tile=int(rng.choice([1, 4, 32]))) as e:
            if i % 2:
                e.add_codes(codes)
            else:
                e.add_synthetic(n, seed=300 + i, kind=1, n_centres=centres, max_flips=flips)
            e.build_index()
            mode = vc.MODE_MIH_APPROX if approx else vc.MODE_MIH_EXACT
            got, cnt, st = e.search_knn(q, k, mode=mode, with_stats=True)
            for r in range(nq):
                ores, ost = mo.find(q[r], k, approximate=approx, use_bitmap=flag == "bitmap", stop_mult=stop_mult)
                g = got[r, : cnt[r]]
                o = np.sort(ores)
                ok = len(g) == len(o) and np.array_equal(g >> SH, o >> SH)
                if ok and len(o):
                    dk = o[-1] >> SH
                    ok = set(g[(g >> SH) < dk].tolist()) == set(o[(o >> SH) < dk].tolist())
                ok = ok and (st[r].radius, st[r].n_sub_reads, st[r].n_candidates) == (ost.radius, ost.n_sub_reads, ost.n_distinct)
                if flag == "bitmap":
                    ok = ok and st[r].n_local_reads == ost.n_local_reads
                ok = ok and np.array_equal(g, canonical(codes, q[r], m, k, ost.radius, signext, id_base))
                if not ok:
                    print("MISMATCH case %d query %d: %s" % (i, r, desc), flush=True)
                    return 1
                checked += 1
            # every seventh case: the same queries repeated into ONE call of 4 100 .. 9 000 (a launch of up to 16 384 slots, the
            # per-slot state laid out for it; the regular cases stay below the 4 096-slot minimum) -- every copy's row, count
            # and statistics must equal the small call's, which the oracle has just checked
            if i % 7 == 5:
                reps = int(rng.integers(4100, 9000)) // nq + 1
                gb, cb, sb = e.search_knn(np.tile(q, (reps, 1)), k, mode=mode, with_stats=True)
                same = np.array_equal(gb.reshape(reps, nq, k), np.broadcast_to(got, (reps, nq, k))) and \
                    np.array_equal(cb.reshape(reps, nq), np.broadcast_to(cnt, (reps, nq)))
                for j in (0, reps // 2, reps - 1):
                    for r in range(nq):
                        a_, b_ = sb[j * nq + r], st[r]
                        same = same and (a_.radius, a_.n_sub_reads, a_.n_local_reads, a_.n_candidates, a_.n_results) == \
                            (b_.radius, b_.n_sub_reads, b_.n_local_reads, b_.n_candidates, b_.n_results)
                if not same:
                    print("BIG-CALL MISMATCH case %d (%d x %d queries): %s" % (i, reps, nq, desc), flush=True)
                    return 1
                checked += reps * nq
                desc += " [call of %d]" % (reps * nq)
            # radius search through both paths against numpy
            rad = int(rng.integers(0, 2 * m + 2))
            a = e.search_radius(q[:4], rad, mode=vc.MODE_MIH_EXACT) if not signext else None
            b = e.search_radius(q[:4], rad, mode=vc.MODE_LINEAR)
            for r in range(min(nq, 4)):
                d = oracle.np_distances(codes, q[r])
                ids = np.nonzero(d <= rad)[0]
                exp = np.sort(oracle.pack(d[ids], ids.astype(np.uint64) + np.uint64(id_base)))
                if not np.array_equal(b[r], exp) or (a is not None and not np.array_equal(a[r], exp)):
                    print("RADIUS MISMATCH case %d query %d radius %d: %s" % (i, r, rad, desc), flush=True)
                    return 1
                checked += 1
        mo.close()
        print("ok case %3d %s (%.0f s)" % (i, desc, time.time() - t0), flush=True)
    print("MIH parity campaign: %d cases, %d query checks bit-exact against the oracle in %.0f s" %
          (n_cases, checked, time.time() - t0), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
