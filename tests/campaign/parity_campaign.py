"""Test infrastructure (GPU box, run by hand; not collected by pytest): a long seeded parity campaign of the linear
verify path against the oracle, beyond what the regular test-suite has time for: databases of several million codes (many chunks per persistent block, threshold
feedback, re-cuts), every code width, ragged sizes, duplicates (ring overflow + recovery), big k, big tiles, and the
selectable kernel shapes.  Prints one line per case and a summary; exit code 1 on the first mismatch.
usage: python tests/campaign/parity_campaign.py [n_cases=120] [seed0=0]   (test infrastructure: imports oracle/)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vc_oracle as oracle  # noqa: E402
from verticut_amd import engine as vc  # noqa: E402

SHAPES = [None, "4,256,3", "4,256,0", "2,256,2", "2,256,3", "1,256,2", "4,512,2", "2,512,3"]


def case(i):
    rng = np.random.default_rng(7000 + i)
    bits = int(rng.choice([64, 128, 128, 256, 512]))
    kind = int(rng.integers(0, 3))          # 0 uniform, 1 clustered, 2 heavy duplicates
    n = int(rng.integers(1, 1 << int(rng.integers(10, 25))))
    k = int(min(n, rng.choice([1, 7, 100, 100, 333, 1000])))
    nq = int(rng.choice([1, 2, 8, 8, 9, 33, 70]))
    qt = int(rng.choice([1, 4, 8, 8, 32, 64]))
    shape = SHAPES[int(rng.integers(0, len(SHAPES)))]
    if shape and bits // 64 * int(shape.split(",")[0]) > 8:
        shape = None
    id_base = int(rng.integers(0, 1 << 20))
    return rng, bits, kind, n, k, nq, qt, shape, id_base


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build(ref=False)
    t_start = time.time()
    checked = 0
    for i in range(seed0, seed0 + n_cases):
        rng, bits, kind, n, k, nq, qt, shape, id_base = case(i)
        if kind == 0:
            codes = oracle.gen_codes(n, bits, 900 + i, first_id=id_base)
        elif kind == 1:
            codes = oracle.gen_codes(n, bits, 900 + i, kind=1, n_centres=int(rng.integers(1, 200)),
                                     max_flips=int(rng.integers(0, 12)), first_id=id_base)
        else:   # a few distinct codes repeated: thousands of ties at the k-th distance (ring overflow + recovery)
            base = oracle.gen_codes(int(rng.integers(1, 6)), bits, 900 + i)
            codes = base[rng.integers(0, len(base), size=n)]
        q = codes[rng.integers(0, n, size=nq)].copy()
        for r in range(nq):
            for b in rng.choice(bits, size=int(rng.integers(0, 6)), replace=False):
                q[r, b // 8] ^= np.uint8(1 << (b % 8))
        if shape:
            os.environ["VC_SCAN_SHAPE"] = shape
        else:
            os.environ.pop("VC_SCAN_SHAPE", None)
        # the verify kernel's cache-resident prefix (plain loads below the boundary, non-temporal above): default 240 MB,
        # none, or a boundary inside the database
        res = (None, None, "0", "1", "4")[i % 5]
        if res is None:
            os.environ.pop("VC_SCAN_RESIDENT_MB", None)
        else:
            os.environ["VC_SCAN_RESIDENT_MB"] = res
        cand_cap = int(rng.choice([0, 0, 4 * k, 4096]))
        with vc.Engine(bits, capacity=n, id_base=id_base, query_tile=qt, cand_cap=cand_cap) as e:
            if kind == 0 and i % 2 == 0:
                e.add_synthetic(n, seed=900 + i)
            else:
                e.add_codes(codes)
            got, cnt = e.search_knn(q, k)
            for r in range(nq):
                exp = oracle.linear_knn(codes, q[r], k, id_base=id_base, threads=8)
                if cnt[r] != len(exp) or not np.array_equal(got[r, : cnt[r]], exp):
                    print("MISMATCH case %d query %d: bits=%d kind=%d n=%d k=%d nq=%d qt=%d shape=%s cap=%d" %
                          (i, r, bits, kind, n, k, nq, qt, shape, cand_cap), flush=True)
                    return 1
                checked += 1
        print("ok case %3d bits=%3d kind=%d n=%8d k=%4d nq=%2d qt=%2d shape=%-8s cap=%5d res=%-4s (%.0f s)" %
              (i, bits, kind, n, k, nq, qt, shape, cand_cap, res, time.time() - t_start), flush=True)
    print("parity campaign: %d cases, %d queries bit-exact against the oracle in %.0f s" %
          (n_cases, checked, time.time() - t_start), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
