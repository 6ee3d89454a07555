"""Dev tool: latency of ONE query per call through the host API (what the reference's drivers do: find() per query,
distributed_image_search.cc:62-85) on 1e8 clustered 128-bit codes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc
n, bits, m, k = 100_000_000, 128, 4, 100
e = vc.Engine(bits, capacity=n, n_tables=m)
e.add_synthetic(n, seed=34, kind=vc.SYNTH_CLUSTERED, n_centres=n // 1000, max_flips=11)
e.build_index()
rng = np.random.default_rng(0)
qs = []
for i in range(200):
    c = e.get_code(int(rng.integers(0, n)))
    for b in rng.choice(bits, size=int(rng.integers(0, 5)), replace=False):
        c[b // 8] ^= np.uint8(1 << (b % 8))
    qs.append(c[None, :].copy())
allq = np.concatenate(qs)
for mode, name in ((vc.MODE_MIH_EXACT, "mih_exact"), (vc.MODE_MIH_APPROX, "mih_approx"), (vc.MODE_LINEAR, "linear")):
    for nq in (1, 16, 128):
        if mode == vc.MODE_LINEAR and nq > 1:
            continue
        batches = [allq[i:i + nq] for i in range(0, len(allq) - nq + 1, nq)][:200]
        for q in batches[:3]:
            e.search_knn(q, k, mode=mode, with_stats=True)
        t0 = time.perf_counter()
        for q in batches:
            e.search_knn(q, k, mode=mode, with_stats=True)
        dt = time.perf_counter() - t0
        print("%s nq=%d: %.1f us per call (%d calls, host API incl. PCIe and statistics)" % (name, nq, dt / len(batches) * 1e6, len(batches)), flush=True)
