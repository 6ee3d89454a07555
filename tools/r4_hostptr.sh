#!/bin/bash
# Dev tool (GPU box): exact top-100 through MIH with HOST pointers (vc_search_knn): pageable numpy buffers vs page-locked ones
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import numpy as np, time, torch
from verticut_amd import engine as vc
n, bits, m, k = 100_000_000, 128, 4, 100
e = vc.Engine(bits, capacity=n, n_tables=m, flags=vc.FLAG_LEAN_TIMING)
e.add_synthetic(n, seed=1, kind=vc.SYNTH_CLUSTERED, n_centres=n//1000, max_flips=11)
e.build_index()
rng = np.random.default_rng(3)
ids = rng.integers(0, n, size=16384)
q = np.stack([e.get_code(int(i)) for i in ids[:2048]])
q = np.tile(q, (8,1))
for Q in (4096, 16384):
    qq = q[:Q].copy()
    for with_stats in (False,):
        e.search_knn(qq, k, mode=vc.MODE_MIH_EXACT, with_stats=with_stats)
        t=time.perf_counter()
        for i in range(4): r = e.search_knn(qq, k, mode=vc.MODE_MIH_EXACT, with_stats=with_stats)
        dt=(time.perf_counter()-t)/4
        print("pageable Q=%d stats=%s: %.3f ms per call = %.2f M q/s" % (Q, with_stats, dt*1e3, Q/dt/1e6))
    # pinned buffers through the raw ABI
    import ctypes as C
    out = torch.empty((Q, k), dtype=torch.int64).pin_memory(); cnt = torch.empty((Q,), dtype=torch.int32).pin_memory()
    qp = torch.from_numpy(qq).pin_memory()
    L = e._L
    def call():
        rc = L.vc_search_knn(e._h, C.c_void_p(qp.data_ptr()), Q, k, vc.MODE_MIH_EXACT, 0, C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr()), None)
        assert rc == 0, rc
    call()
    t=time.perf_counter()
    for i in range(4): call()
    dt=(time.perf_counter()-t)/4
    print("pinned   Q=%d: %.3f ms per call = %.2f M q/s" % (Q, dt*1e3, Q/dt/1e6))
    ref = e.search_knn(qq, k, mode=vc.MODE_MIH_EXACT)[0]
    print("  rows equal:", bool(np.array_equal(out.numpy().view(np.uint64), ref)))
P
