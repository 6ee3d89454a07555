"""Dev tool: one engine, fixed n/bits/qt, a few timed searches (target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc
n = int(float(sys.argv[1])); bits = int(sys.argv[2]); qt = int(sys.argv[3]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
e = vc.Engine(bits, capacity=n, query_tile=qt)
e.add_synthetic(n, seed=34)
q = np.random.default_rng(0).integers(0, 256, size=(qt, bits // 8), dtype=np.uint8)
for _ in range(reps):
    e.search_knn(q, 100)
    t = e.timing()
    print(f"n={n} qt={qt} scan_ms={t.scan_ms:.3f} total_ms={t.total_ms:.3f} GB/s={t.scan_bytes/t.scan_ms/1e6:.1f}", flush=True)
