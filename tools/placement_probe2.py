"""Dev tool: does the allocation lottery of the streaming rate need two concurrent column streams?  Engines with one
64-bit column of 2e9 codes (16 GB, one stream) and with two columns of 1e9 128-bit codes (16 GB, two streams 8 GB apart)
alternate in one process; qt = 1 scan time of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc

rng = np.random.default_rng(0)
engines = []
for i in range(8):
    bits, n = ((64, 2 * 10**9) if i % 2 == 0 else (128, 10**9))
    e = vc.Engine(bits, capacity=n, query_tile=8)
    e.add_synthetic(n, seed=34)
    engines.append((bits, n, e))

def sample(e, q, reps=12):
    for _ in range(3):
        e.search_knn(q, 100)
    e.timing()
    for _ in range(reps):
        e.search_knn(q, 100)
    t = e.timing()
    return t.scan_ms / t.scan_launches

for rnd in range(2):
    for bits in (64, 128):
        q = rng.integers(0, 256, size=(1, bits // 8), dtype=np.uint8)
        print("round %d, %3d-bit (%d column%s), qt=1: " % (rnd, bits, bits // 64, "s" if bits > 64 else "") +
              "  ".join("%.3f" % sample(e, q) for b, n, e in engines if b == bits), flush=True)
