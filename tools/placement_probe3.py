"""Dev tool: which alignment of the column stride / of the allocation size removes the streaming-rate lottery?  Engines
hold the same 1e9 codes; knobs VC_STRIDE_ALIGN (items) and VC_ALLOC_ALIGN (bytes) are set per engine at creation;
three engines of each kind, interleaved, one process; GB/s at qt=1 and qt=8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc

n = int(os.environ.get('PROBE_N', 10**9))
kinds = [("stride 64 KiB, alloc exact", None, None), ("stride 64 KiB, alloc 1 GiB", None, 1 << 30),
         ("stride 256 MiB", 1 << 25, None), ("stride 512 MiB", 1 << 26, None), ("stride 1 GiB", 1 << 27, None)]
if os.environ.get("PROBE_KINDS"):
    kinds = [k for k in kinds if k[0] in os.environ["PROBE_KINDS"].split(";")]
rng = np.random.default_rng(0)
q = rng.integers(0, 256, size=(8, 16), dtype=np.uint8)
engines = []
for rep in range(3):
    for name, sa, aa in kinds:
        for k, v in (("VC_STRIDE_ALIGN", sa), ("VC_ALLOC_ALIGN", aa)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        e = vc.Engine(128, capacity=n, query_tile=8)
        e.add_synthetic(n, seed=34)
        engines.append((name, e))

def rate(e, qq):
    for _ in range(3):
        e.search_knn(qq, 100)
    e.timing()
    for _ in range(10):
        e.search_knn(qq, 100)
    t = e.timing()
    return n * 16 / (t.scan_ms / t.scan_launches) / 1e6

for qt in (1, 8):
    for name, sa, aa in kinds:
        print("qt=%d %-28s GB/s " % (qt, name) + "  ".join("%.0f" % rate(e, q[:qt]) for nm, e in engines if nm == name), flush=True)
