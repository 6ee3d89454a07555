"""Dev tool (GPU box): index build time (the step before the path) -- synthetic codes resident
in HBM -> sorted id runs + occupancy bitmaps + rank directories of every table (vc_build_index: key extraction, the
hand-written stable radix sort of vc_sort.hip, run heads, scans).
usage: python tools/bench_build.py [n=1e9] [bits=128] [tables=4]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import verticut_amd.engine as vc  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 9
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = int(sys.argv[3]) if len(sys.argv) > 3 else 4
with vc.Engine(bits, capacity=n, n_tables=m) as e:
    t0 = time.perf_counter()
    e.add_synthetic(n, seed=34)
    t1 = time.perf_counter()
    e.build_index()
    t2 = time.perf_counter()
    print("n = %d, %d-bit codes, %d tables of %d-bit substrings: fill %.3f s, index build %.3f s (%.1f M codes/s, %.1f M (key, id) pairs sorted per second)"
          % (n, bits, m, bits // m, t1 - t0, t2 - t1, n / (t2 - t1) / 1e6, n * m / (t2 - t1) / 1e6))
    t3 = time.perf_counter()
    e.build_index()
    print("second build (buffers warm): %.3f s" % (time.perf_counter() - t3))
