"""Dev tool: single-column (64-bit) databases: does the allocation size decide the streaming rate?  2e9 codes in engines
created with capacity 2e9 (16.0e9 bytes) and with capacity 2^31 (16 GiB exactly), interleaved in one process; GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc

n = 2 * 10**9
rng = np.random.default_rng(0)
q = rng.integers(0, 256, size=(8, 8), dtype=np.uint8)
engines = []
for i in range(8):
    cap = n if i % 2 == 0 else 1 << 31
    e = vc.Engine(64, capacity=cap, query_tile=8)
    e.add_synthetic(n, seed=34)
    engines.append((cap, e))

def rate(e, qq):
    for _ in range(3):
        e.search_knn(qq, 100)
    e.timing()
    for _ in range(10):
        e.search_knn(qq, 100)
    t = e.timing()
    return n * 8 / (t.scan_ms / t.scan_launches) / 1e6

for qt in (1, 8):
    for cap in (n, 1 << 31):
        print("qt=%d capacity %10d: GB/s " % (qt, cap) + "  ".join("%.0f" % rate(e, q[:qt]) for c, e in engines if c == cap), flush=True)
