"""Generate tests/golden/search_fixtures.json from the ORACLE (oracle/vc_oracle.cc), not from the reference:
the reference's search loops cannot be built in this image, so these vectors pin
"GPU == oracle == what was committed", while the primitives underneath are pinned to the reference by
primitives.json.  Databases are regenerated from (seed, kind, ...) by the shared counter-based generator.

    python tests/golden/make_search_fixtures.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vc_oracle as vo  # noqa: E402

SH = np.uint64(32)
GRID = [
    # bits, m, n, kind, centres, flips, k
    (128, 4, 4096, 0, 0, 0, 10),
    (128, 4, 65536, 1, 300, 10, 100),
    (128, 4, 65536, 1, 300, 10, 1),
    (64, 4, 65536, 1, 200, 5, 10),
    (64, 2, 40000, 1, 200, 6, 10),
    (256, 8, 30000, 1, 150, 14, 100),
]
out = {"source": "oracle/vc_oracle.cc (restatement of linear_search.cc:39-64 and search_worker.cc:65-264)", "fixtures": []}
for bits, m, n, kind, centres, flips, k in GRID:
    rng = np.random.default_rng(bits * 1000 + m * 10 + k)
    codes = vo.gen_codes(n, bits, 34, kind, centres, flips)
    q = codes[rng.integers(0, n, size=4)].copy()
    for i in range(4):
        for _ in range(i):                      # 0..3 bit flips
            b = int(rng.integers(0, bits))
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    fx = {"bits": bits, "m": m, "n": n, "seed": 34, "kind": kind, "n_centres": centres, "max_flips": flips, "k": k,
          "queries": [bytes(x).hex() for x in q], "linear": [], "mih_exact": [], "mih_approx": []}
    mo = vo.MihOracle(codes, m, key_mode=1) if kind == 1 else None
    for i in range(4):
        fx["linear"].append([int(v) for v in vo.linear_knn(codes, q[i], k)])
        if mo is not None:
            stop_mult = min(m, 4)
            res, st = mo.find(q[i], k, stop_mult=stop_mult)
            mind = vo.np_sub_distances(codes, q[i], m).min(axis=1)
            d = vo.np_distances(codes, q[i])
            seen = mind <= st.radius
            can = np.sort(vo.pack(d[seen], np.nonzero(seen)[0].astype(np.uint64)))[:k]
            assert np.array_equal(np.sort(res >> SH), can >> SH)
            fx["mih_exact"].append({"radius": int(st.radius), "n_sub_reads": int(st.n_sub_reads),
                                    "n_candidates": int(st.n_distinct), "result": [int(v) for v in can]})
            ares, ast = mo.find(q[i], k, approximate=True)
            seen = mind <= ast.radius
            acan = np.sort(vo.pack(d[seen], np.nonzero(seen)[0].astype(np.uint64)))[:k]
            assert np.array_equal(np.sort(ares >> SH), acan >> SH)
            fx["mih_approx"].append({"radius": int(ast.radius), "n_candidates": int(ast.n_distinct),
                                     "result": [int(v) for v in acan]})
    out["fixtures"].append(fx)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "search_fixtures.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print(path, os.path.getsize(path))
