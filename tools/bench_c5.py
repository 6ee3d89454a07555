"""Dev tool: BASELINE config 5 shape on ONE GPU's share: 256-bit codes, 5e8 codes (16 GB), 4096 queries per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc
n, bits, nq, k = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000, 256, 4096, 100
rng = np.random.default_rng(5)
q = rng.integers(0, 256, size=(nq, bits // 8), dtype=np.uint8)
for qt in (32, 256, 4096):
    e = vc.Engine(bits, capacity=n, query_tile=qt)
    e.add_synthetic(n, seed=34)
    e.search_knn(q[:qt], k)
    e.timing()
    t0 = time.perf_counter(); out, cnt = e.search_knn(q, k); dt = time.perf_counter() - t0
    t = e.timing()
    print(f"c5 n={n} bits={bits} nq={nq} query_tile={qt}: {nq/dt:8.1f} qps wall ({dt*1e3:.1f} ms), scan launches {t.scan_launches}, "
          f"scan {t.scan_ms:.1f} ms, algorithmic {t.scan_bytes/t.scan_ms/1e6:.0f} GB/s, pairs/s {nq*n/(t.scan_ms*1e-3):.3e}", flush=True)
    e.close()
