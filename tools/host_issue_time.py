import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from verticut_amd import engine as vc
e = vc.Engine(128, capacity=125_000_000, query_tile=8, flags=vc.FLAG_LEAN_TIMING)
e.add_synthetic(125_000_000, seed=3)
dq = torch.randint(0, 256, (8, 16), dtype=torch.uint8, device="cuda")
out = torch.empty((8, 100), dtype=torch.int64, device="cuda"); cnt = torch.empty((8,), dtype=torch.int32, device="cuda")
for _ in range(20): e.search_knn_dev(dq.data_ptr(), 8, 100, out.data_ptr(), cnt.data_ptr(), stream=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): e.search_knn_dev(dq.data_ptr(), 8, 100, out.data_ptr(), cnt.data_ptr(), stream=None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per call: %.1f us; GPU-complete per call: %.1f us" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
