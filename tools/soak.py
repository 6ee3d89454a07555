"""Dev tool: create/destroy churn and a long run of device-API calls; prints device memory in use along the way."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from verticut_amd import engine as vc

def used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20

torch.cuda.init()
rng = np.random.default_rng(1)
base = used_mb()
for r in range(30):
    with vc.Engine(128, capacity=2_000_000, n_tables=4, query_tile=8) as e:
        e.add_synthetic(2_000_000, seed=r)
        e.build_index()
        q = rng.integers(0, 256, size=(20, 16), dtype=np.uint8)
        e.search_knn(q, 50)
        e.search_knn(q, 50, mode=vc.MODE_MIH_EXACT)
        e.search_radius(q[:4], 20)
    if r % 10 == 9:
        print("after %2d create/destroy cycles: %+.1f MiB vs start" % (r + 1, used_mb() - base), flush=True)
with vc.Engine(128, capacity=20_000_000, query_tile=8, flags=vc.FLAG_LEAN_TIMING) as e:
    e.add_synthetic(20_000_000, seed=3)
    dq = torch.from_numpy(rng.integers(0, 256, size=(8, 16), dtype=np.uint8)).cuda()
    out = torch.empty((8, 100), dtype=torch.int64, device="cuda")
    cnt = torch.empty((8,), dtype=torch.int32, device="cuda")
    e.search_knn_dev(dq.data_ptr(), 8, 100, out.data_ptr(), cnt.data_ptr(), stream=None)
    torch.cuda.synchronize()
    ref = out.clone()
    m0 = used_mb()
    for i in range(20000):
        e.search_knn_dev(dq.data_ptr(), 8, 100, out.data_ptr(), cnt.data_ptr(), stream=None)
        if i % 5000 == 4999:
            torch.cuda.synchronize()
            t = e.timing()
            assert torch.equal(out, ref)
            print("after %5d device-API calls: %+.1f MiB, timed scans since last report %d" % (i + 1, used_mb() - m0, t.scan_launches), flush=True)
print("soak ok")
