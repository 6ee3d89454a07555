"""Dev tool: sweep query-tile size / grid of the verify kernel (prints GB/s against algorithmic bytes).
usage: sweep_scan.py N BITS TILES BLOCKS   (VC_SCAN_SHAPE=U,BLK,DB env picks the kernel shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tiles = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,12,16,24,32,64").split(",")]
blocks = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
k = 100
rng = np.random.default_rng(0)
shape = os.environ.get("VC_SCAN_SHAPE", "default")
for sb in blocks:
    e = vc.Engine(bits, capacity=n, query_tile=max(tiles), scan_blocks=sb)
    e.add_synthetic(n, seed=34)
    for qt in tiles:
        q = rng.integers(0, 256, size=(qt, bits // 8), dtype=np.uint8)
        e.search_knn(q, k)
        e.timing()
        ms = []
        for _ in range(5):
            e.search_knn(q, k)
            t = e.timing()
            ms.append((t.scan_ms, t.total_ms))
        s = sorted(ms)[len(ms) // 2]
        gb = n * bits / 8 / 1e9
        print(f"shape={shape} n={n} bits={bits} blocks={sb} qt={qt:4d} scan_ms={s[0]:8.3f} total_ms={s[1]:8.3f} "
              f"scan_GBps={gb / s[0] * 1e3:8.1f} qps={qt / s[1] * 1e3:10.1f}", flush=True)
    e.close()
