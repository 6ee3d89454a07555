"""Dev tool: is the run-to-run spread of the verify kernel tied to where the buffers land?  Several engines (each with
its own copy of the same 16 GB database) live in ONE process; their scan times are sampled round-robin, at qt = 1
(pure streaming) and qt = 8; then the small per-query state of one engine is reallocated (a bigger ring) and timed again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from verticut_amd import engine as vc

n, bits, k = 10**9, 128, 100
rng = np.random.default_rng(0)
q8 = rng.integers(0, 256, size=(8, bits // 8), dtype=np.uint8)
engines = []
for i in range(5):
    e = vc.Engine(bits, capacity=n, query_tile=16)
    e.add_synthetic(n, seed=34)
    engines.append(e)

def sample(e, q, kk=k, reps=12):
    for _ in range(3):
        e.search_knn(q, kk)
    e.timing()
    for _ in range(reps):
        e.search_knn(q, kk)
    t = e.timing()
    return t.scan_ms / t.scan_launches

for rnd in range(2):
    print("qt=1 round %d: " % rnd + "  ".join("%.3f" % sample(e, q8[:1]) for e in engines), flush=True)
for rnd in range(2):
    print("qt=8 round %d: " % rnd + "  ".join("%.3f" % sample(e, q8) for e in engines), flush=True)
# same databases, new small buffers: a 200-query call (tiles of 8, groups of 64) regrows the per-query state
# (counters, histograms, thresholds) and the ring at new addresses; the engines then time 8 queries again
q200 = rng.integers(0, 256, size=(200, bits // 8), dtype=np.uint8)
for e in engines:
    e.search_knn(q200, k)
print("qt=8 after the state buffers moved: " + "  ".join("%.3f" % sample(e, q8) for e in engines), flush=True)
print("qt=8 once more:                     " + "  ".join("%.3f" % sample(e, q8) for e in engines), flush=True)
print("qt=16:                              " + "  ".join("%.3f" % sample(e, np.concatenate([q8, q8 ^ 0x5A])[:16]) for e in engines), flush=True)
