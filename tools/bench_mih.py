"""Dev tool: MIH path timings at BASELINE config-2 scale (64-bit codes, 1e8 DB, radius-8 neighbour search) and
exact k-NN on clustered 128-bit data.  Prints wall-clock QPS of the host API (includes PCIe for queries/results)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc

def near_queries(e, n, nq, bits, max_flips, rng):
    q = np.empty((nq, bits // 8), dtype=np.uint8)
    for i in range(nq):
        c = e.get_code(int(rng.integers(0, n)))
        for b in rng.choice(bits, size=int(rng.integers(0, max_flips + 1)), replace=False):
            c[b // 8] ^= np.uint8(1 << (b % 8))
        q[i] = c
    return q

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
rng = np.random.default_rng(0)
if which == "c2":
    n, bits, nq, radius = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000, 64, 1024, 8
    for m in (2, 4):
        e = vc.Engine(bits, capacity=n, n_tables=m)
        e.add_synthetic(n, seed=34)
        t0 = time.perf_counter(); e.build_index(); tb = time.perf_counter() - t0
        q = near_queries(e, n, nq, bits, radius, rng)
        modes = ((vc.MODE_MIH_EXACT, "mih"),) if os.environ.get("MIH_ONLY") else ((vc.MODE_MIH_EXACT, "mih"), (vc.MODE_LINEAR, "linear"))
        for mode, name in modes:
            e.search_radius(q, radius, mode=mode)   # warm: buffers sized for the full batch
            t0 = time.perf_counter(); res = e.search_radius(q, radius, mode=mode); dt = time.perf_counter() - t0
            print(f"c2 n={n} bits={bits} m={m} s={bits//m} radius={radius} {name}: build={tb:.2f}s {nq/dt:9.1f} qps "
                  f"({dt*1e3:.1f} ms / {nq} queries, mean hits {np.mean([len(r) for r in res]):.2f})", flush=True)
        e.close()
else:
    n, bits, m, k, nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000, 128, 4, 100, 256
    e = vc.Engine(bits, capacity=n, n_tables=m)
    e.add_synthetic(n, seed=34, kind=vc.SYNTH_CLUSTERED, n_centres=max(n // 1000, 1), max_flips=11)
    t0 = time.perf_counter(); e.build_index(); tb = time.perf_counter() - t0
    q = near_queries(e, n, nq, bits, 4, rng)
    for mode, name in ((vc.MODE_MIH_EXACT, "mih_exact"), (vc.MODE_MIH_APPROX, "mih_approx"), (vc.MODE_LINEAR, "linear")):
        e.search_knn(q, k, mode=mode)   # warm: buffers sized for the full batch
        t0 = time.perf_counter(); out, cnt, st = e.search_knn(q, k, mode=mode, with_stats=True); dt = time.perf_counter() - t0
        print(f"knn n={n} clustered bits={bits} m={m} k={k} {name}: build={tb:.2f}s {nq/dt:9.1f} qps ({dt*1e3:.1f} ms / {nq}), "
              f"mean radius {np.mean([s.radius for s in st]):.2f}, mean candidates {np.mean([s.n_candidates for s in st]):.0f}", flush=True)
    e.close()
