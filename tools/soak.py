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
# exact MIH k-NN and radius search through the device API: the host learns each call's counters / total by polling mapped
# memory (sequence numbers); 20 000 calls with alternating batches must keep returning each batch's own rows
with vc.Engine(128, capacity=5_000_000, n_tables=4) as e:
    e.add_synthetic(5_000_000, seed=7, kind=vc.SYNTH_CLUSTERED, n_centres=5000, max_flips=9)
    e.build_index()
    nq, k = 512, 20
    hq = []
    for _ in range(2):                              # near queries: database rows with a flipped bit
        h = np.stack([e.get_code(int(rng.integers(0, 5_000_000))) for _ in range(nq)])
        h[np.arange(nq), rng.integers(0, 16, size=nq)] ^= (np.uint8(1) << rng.integers(0, 8, size=nq).astype(np.uint8))
        hq.append(h)
    dq = [torch.from_numpy(h).cuda() for h in hq]
    out = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    rout = torch.empty((nq * 64,), dtype=torch.int64, device="cuda")
    roff = torch.empty((nq + 1,), dtype=torch.int64, device="cuda")
    ref, rref = [], []
    for b in range(2):
        e.search_knn_dev(dq[b].data_ptr(), nq, k, out.data_ptr(), cnt.data_ptr(), mode=vc.MODE_MIH_EXACT, stream=None)
        torch.cuda.synchronize()
        ref.append((out.clone(), cnt.clone()))
        e.search_radius_dev(dq[b].data_ptr(), nq, 6, rout.data_ptr(), rout.numel(), roff.data_ptr(), mode=vc.MODE_MIH_EXACT, stream=None)
        torch.cuda.synchronize()
        rref.append((rout[: int(roff[nq])].clone(), roff.clone()))
    for i in range(20000):
        b = i & 1
        e.search_knn_dev(dq[b].data_ptr(), nq, k, out.data_ptr(), cnt.data_ptr(), mode=vc.MODE_MIH_EXACT, stream=None)
        if i % 997 == 0:
            torch.cuda.synchronize()
            assert torch.equal(out, ref[b][0]) and torch.equal(cnt, ref[b][1]), i
        e.search_radius_dev(dq[b].data_ptr(), nq, 6, rout.data_ptr(), rout.numel(), roff.data_ptr(), mode=vc.MODE_MIH_EXACT, stream=None)
        if i % 997 == 0:
            torch.cuda.synchronize()
            assert torch.equal(roff, rref[b][1]) and torch.equal(rout[: int(roff[nq])], rref[b][0]), i
        if i % 5000 == 4999:
            print("after %5d MIH k-NN + radius device-API calls: ok" % (i + 1), flush=True)
print("soak ok")
