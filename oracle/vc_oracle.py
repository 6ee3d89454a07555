"""ctypes loader for the CPU oracle (oracle/vc_oracle.cc) and, when built, the reference
veneer (oracle/_ref/libvcref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (verticut_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvcoracle.so")
_REF = os.path.join(_HERE, "_ref", "libvcref.so")


def build(ref=True):
    """Compile the oracle (always) and the reference veneer (only where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class FindStats(C.Structure):
    _fields_ = [
        ("radius", C.c_uint32),
        ("n_results", C.c_uint32),
        ("n_main_reads", C.c_uint64),
        ("n_sub_reads", C.c_uint64),
        ("n_local_reads", C.c_uint64),
        ("n_sub_reads_all", C.c_uint64),
        ("n_local_reads_all", C.c_uint64),
        ("n_candidates", C.c_uint64),
        ("n_distinct", C.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build(ref=False)
        L = C.CDLL(_LIB)
        u8p = C.c_void_p
        L.vco_hamming.restype = C.c_int
        L.vco_hamming.argtypes = [u8p, u8p, C.c_size_t]
        L.vco_binary_to_int.restype = C.c_uint32
        L.vco_binary_to_int.argtypes = [C.c_char_p, C.c_int]
        L.vco_bitmap_get.restype = C.c_int
        L.vco_bitmap_get.argtypes = [u8p, C.c_uint64]
        L.vco_bitmap_set.argtypes = [u8p, C.c_uint64]
        L.vco_bitmap_reset.argtypes = [u8p, C.c_uint64]
        L.vco_gen_codes.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                    C.c_uint32, C.c_uint32]
        for f in (L.vco_linear_knn_ref, L.vco_linear_knn):
            f.restype = C.c_uint32
            f.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint32, C.c_uint32, u8p]
        L.vco_linear_knn_mt.restype = C.c_uint32
        L.vco_linear_knn_mt.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.vco_pool_create.restype = C.c_void_p
        L.vco_pool_create.argtypes = [C.c_uint32]
        L.vco_pool_destroy.argtypes = [C.c_void_p]
        L.vco_pool_size.restype = C.c_uint32
        L.vco_pool_size.argtypes = [C.c_void_p]
        L.vco_gen_codes_pool.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                         C.c_uint32, C.c_uint32]
        L.vco_linear_knn_pool.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          u8p, u8p]
        L.vco_linear_radius_pool.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             u8p, C.c_uint64, u8p]
        L.vco_mih_create.restype = C.c_void_p
        L.vco_mih_create.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.vco_mih_destroy.argtypes = [C.c_void_p]
        L.vco_mih_bucket.restype = C.c_uint32
        L.vco_mih_bucket.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u8p, C.c_uint32]
        L.vco_mih_key.restype = C.c_uint32
        L.vco_mih_key.argtypes = [C.c_void_p, u8p, C.c_uint32]
        L.vco_mih_find.restype = C.c_uint32
        L.vco_mih_find.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, u8p,
                                   C.POINTER(FindStats)]
        L.vco_mih_find_mt.restype = C.c_uint32
        L.vco_mih_find_mt.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, u8p,
                                      C.POINTER(FindStats)]
        L.vco_mih_radius.restype = C.c_uint64
        L.vco_mih_radius.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_uint32, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


_ref = None


def ref():
    """The reference's own compute_hamming_dist / binaryToInt / ImageBitmap, or None if not built."""
    global _ref
    if _ref is None and os.path.exists(_REF):
        R = C.CDLL(_REF)
        R.vcref_hamming.restype = C.c_int
        R.vcref_hamming.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        R.vcref_binary_to_int.restype = C.c_uint32
        R.vcref_binary_to_int.argtypes = [C.c_char_p, C.c_int]
        R.vcref_bitmap_new.restype = C.c_void_p
        R.vcref_bitmap_new.argtypes = [C.c_ulong]
        R.vcref_bitmap_free.argtypes = [C.c_void_p]
        R.vcref_bitmap_set.argtypes = [C.c_void_p, C.c_ulong]
        R.vcref_bitmap_reset.argtypes = [C.c_void_p, C.c_ulong]
        R.vcref_bitmap_get.restype = C.c_int
        R.vcref_bitmap_get.argtypes = [C.c_void_p, C.c_ulong]
        R.vcref_bitmap_data.restype = C.c_void_p
        R.vcref_bitmap_data.argtypes = [C.c_void_p]
        _ref = R
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bytes(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.uint8
    return a


# ------------------------------------------------------------------ primitives
def hamming(a, b):
    a, b = _bytes(a), _bytes(b)
    return lib().vco_hamming(_p(a), _p(b), a.size)


def binary_to_int(raw: bytes, length=None):
    return lib().vco_binary_to_int(raw, len(raw) if length is None else length)


def gen_codes(n, bits, seed, kind=0, n_centres=0, max_flips=0, first_id=0):
    """Synthetic DB (row-major uint8 [n, bits/8]); same definition as the HIP generator."""
    out = np.empty((n, bits // 8), dtype=np.uint8)
    lib().vco_gen_codes(_p(out), first_id, n, bits, seed, kind, n_centres, max_flips)
    return out


# ------------------------------------------------------------------ linear scan
def linear_knn_ref(codes, query, k, id_base=0):
    """linear_search.cc:39-64 order (farthest first); returns packed uint64 array."""
    codes, query = _bytes(codes), _bytes(query)
    out = np.empty(max(k, 1), dtype=np.uint64)
    c = lib().vco_linear_knn_ref(_p(codes), codes.shape[0], codes.shape[1], _p(query), k, id_base, _p(out))
    return out[:c].copy()


def linear_knn(codes, query, k, id_base=0, threads=1):
    """Canonical contract: the k smallest packed dist<<32|id, ascending."""
    codes, query = _bytes(codes), _bytes(query)
    out = np.empty(max(k, 1), dtype=np.uint64)
    if threads > 1:
        c = lib().vco_linear_knn_mt(_p(codes), codes.shape[0], codes.shape[1], _p(query), k, id_base, threads, _p(out))
    else:
        c = lib().vco_linear_knn(_p(codes), codes.shape[0], codes.shape[1], _p(query), k, id_base, _p(out))
    return out[:c].copy()


# ------------------------------------------------------------------ big databases: a persistent worker pool
def host_threads():
    """threads this process may really use: the affinity mask, capped by the cgroup's CPU quota (the GPU box reports 256
    cores and an affinity mask of 256, but a one-GPU job's cpu.max is 16 CPUs: 256 threads would share 16 cores' time)"""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], int(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    period = int(g.read())
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(quota) // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


class Pool:
    """Worker threads created once (oracle/vc_oracle.cc VcoPool): the full-size parity tests and bench.py's all-cores
    CPU baseline run the linear_search.cc scan through it, slab by slab."""

    def __init__(self, threads=None):
        self.h = lib().vco_pool_create(threads or host_threads())
        self.threads = lib().vco_pool_size(self.h)

    def close(self):
        if self.h:
            lib().vco_pool_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def gen_codes(self, n, bits, seed, kind=0, n_centres=0, max_flips=0, first_id=0, out=None):
        if out is None:
            out = np.empty((n, bits // 8), dtype=np.uint8)
        lib().vco_gen_codes_pool(self.h, _p(out), first_id, n, bits, seed, kind, n_centres, max_flips)
        return out[:n]

    def linear_knn(self, codes, queries, k, id_base=0):
        """canonical top-k (ascending packed, padded with ~0) of every query over one slab: ([nq, k], counts[nq])"""
        codes, queries = _bytes(codes), _bytes(queries)
        nq = queries.shape[0]
        out = np.empty((nq, k), dtype=np.uint64)
        cnt = np.empty(nq, dtype=np.uint32)
        lib().vco_linear_knn_pool(self.h, _p(codes), codes.shape[0], codes.shape[1], _p(queries), nq, k, id_base, _p(out), _p(cnt))
        return out, cnt

    def linear_radius(self, codes, queries, radius, id_base=0, cap=4096):
        """every item within `radius` of every query over one slab: list of ascending packed arrays"""
        codes, queries = _bytes(codes), _bytes(queries)
        nq = queries.shape[0]
        while True:
            out = np.empty((nq, cap), dtype=np.uint64)
            cnt = np.empty(nq, dtype=np.uint64)
            lib().vco_linear_radius_pool(self.h, _p(codes), codes.shape[0], codes.shape[1], _p(queries), nq, radius, id_base,
                                         _p(out), cap, _p(cnt))
            if int(cnt.max(initial=0)) <= cap:
                return [out[i, : int(cnt[i])].copy() for i in range(nq)]
            cap = int(cnt.max())


def slabs(n, slab):
    """[lo, hi) ranges covering 0..n"""
    return [(lo, min(lo + slab, n)) for lo in range(0, n, slab)]


def linear_knn_slabbed(pool, n, bits, seed, queries, k, slab=1 << 27, kind=0, n_centres=0, max_flips=0):
    """linear_search.cc:39-64 over the synthetic database of `n` codes WITHOUT holding it: generated slab by slab
    (same definition as the device generator), scanned for all queries, per-slab top-k merged.  [nq, k] ascending."""
    queries = _bytes(queries)
    best = None
    buf = np.empty((min(slab, n), bits // 8), dtype=np.uint8)
    for lo, hi in slabs(n, slab):
        codes = pool.gen_codes(hi - lo, bits, seed, kind, n_centres, max_flips, first_id=lo, out=buf)
        part, _ = pool.linear_knn(codes, queries, k, id_base=lo)
        best = part if best is None else np.sort(np.concatenate([best, part], axis=1), axis=1)[:, :k]
    return best


def linear_radius_slabbed(pool, n, bits, seed, queries, radius, slab=1 << 27):
    queries = _bytes(queries)
    res = [np.empty(0, dtype=np.uint64) for _ in range(queries.shape[0])]
    buf = np.empty((min(slab, n), bits // 8), dtype=np.uint8)
    for lo, hi in slabs(n, slab):
        codes = pool.gen_codes(hi - lo, bits, seed, first_id=lo, out=buf)
        part = pool.linear_radius(codes, queries, radius, id_base=lo)
        res = [np.concatenate([a, b]) for a, b in zip(res, part)]     # slabs ascend in id, rows are sorted by (dist, id)
    return [np.sort(r) for r in res]


# ------------------------------------------------------------------ MIH
class MihOracle:
    """SearchWorker (search_worker.cc) over in-memory buckets built by rule a12."""

    def __init__(self, codes, m, key_mode=0, id_base=0):
        codes = _bytes(codes)
        self.codes = codes
        self.m = m
        self.h = lib().vco_mih_create(_p(codes), codes.shape[0], codes.shape[1], m, key_mode, id_base)
        if not self.h:
            raise ValueError("bad (nbytes, m)")

    def close(self):
        if self.h:
            lib().vco_mih_destroy(self.h)
            self.h = None

    __del__ = close

    def key(self, code, table):
        code = _bytes(code)
        return lib().vco_mih_key(self.h, _p(code), table)

    def bucket(self, table, index, cap=1 << 20):
        ids = np.empty(cap, dtype=np.uint32)
        n = lib().vco_mih_bucket(self.h, table, index, _p(ids), cap)
        return ids[: min(n, cap)].copy()

    def find(self, query, k, approximate=False, use_bitmap=False, stop_mult=4, threads=1):
        """Returns (packed farthest-first, FindStats).  threads = ranks run concurrently (mpirun -n m); same result."""
        query = _bytes(query)
        out = np.empty(max(k, 1) * (1 if not approximate else 1) + 8, dtype=np.uint64)
        st = FindStats()
        c = lib().vco_mih_find_mt(self.h, _p(query), k, int(approximate), int(use_bitmap), stop_mult, threads, _p(out),
                                  C.byref(st))
        return out[:c].copy(), st

    def radius(self, query, radius, threads=1, cap=1 << 16):
        """all items within full distance <= radius (search_R_neighbors shells 0..radius/m per rank, gather, dedup):
        (packed ascending, number of bucket gets issued over all ranks)"""
        query = _bytes(query)
        out = np.empty(cap, dtype=np.uint64)
        probes = C.c_uint64()
        c = lib().vco_mih_radius(self.h, _p(query), radius, threads, _p(out), cap, C.byref(probes))
        if c > cap:
            return self.radius(query, radius, threads, cap=int(c))
        return out[:c].copy(), probes.value


# ------------------------------------------------------------------ numpy helpers (contract checks)
def np_distances(codes, query):
    """Full Hamming distances of every row to query (numpy; independent of the C code)."""
    x = np.bitwise_xor(_bytes(codes), _bytes(query)[None, :])
    return np.unpackbits(x, axis=1).sum(axis=1).astype(np.uint32)


def np_sub_distances(codes, query, m):
    """[n, m] per-substring distances (substring t = bytes [t*nlb, (t+1)*nlb))."""
    codes = _bytes(codes)
    n, nb = codes.shape
    x = np.bitwise_xor(codes, _bytes(query)[None, :])
    bits = np.unpackbits(x, axis=1).reshape(n, m, -1)
    return bits.sum(axis=2).astype(np.uint32)


def pack(dist, ids):
    return (dist.astype(np.uint64) << np.uint64(32)) | ids.astype(np.uint64)
