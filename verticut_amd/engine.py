"""ctypes binding of include/verticut_gpu.h -- the only way Python reaches the kernels.

There is no Python or CPU implementation of any search here: if libverticut_gpu.so is missing
or no gfx950 device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VERTICUT_GPU_LIB") or os.path.join(_HERE, "lib", "libverticut_gpu.so")   # env: A/B builds

VC_ABI_VERSION = 2
VC_OK, VC_NOT_FOUND = 0, 1
VC_ERR_INVALID, VC_ERR_NO_DEVICE, VC_ERR_HIP, VC_ERR_NOMEM, VC_ERR_STATE, VC_ERR_CAPACITY = -1, -2, -3, -4, -5, -6
MODE_LINEAR, MODE_MIH_EXACT, MODE_MIH_APPROX = 0, 1, 2
FLAG_USE_BITMAP, FLAG_REF_SIGNEXT_KEYS, FLAG_REF_STOP_LITERAL4, FLAG_LEAN_TIMING = 1, 2, 4, 8
SYNTH_UNIFORM, SYNTH_CLUSTERED = 0, 1
ORDER_ASCENDING, ORDER_FARTHEST_FIRST = 0, 1
PACK_INF = np.uint64(0xFFFFFFFFFFFFFFFF)
STREAM_OWN = C.c_void_p(-1)   # VC_STREAM_OWN; None / 0 = the HIP null stream (PyTorch's default stream)

# every symbol include/verticut_gpu.h declares (tests check the .so exports exactly these)
EXPORTS = [
    "vc_create", "vc_destroy", "vc_last_error", "vc_strerror", "vc_abi_version", "vc_add_codes", "vc_add_synthetic",
    "vc_size", "vc_get_code", "vc_build_index", "vc_get_bucket", "vc_bitmap_test", "vc_bitmap_read", "vc_search_knn",
    "vc_search_knn_dev", "vc_search_radius", "vc_merge_topk_dev", "vc_get_timing", "vc_set_stream",
    "vc_load_code_file", "vc_save_code_file", "vc_write_bitmap_file", "vc_device_status",
    "vc_search_radius_dev", "vc_read_bitmap_file", "vc_save_index", "vc_load_index",
    "vc_sharded_create", "vc_sharded_destroy", "vc_sharded_last_error", "vc_sharded_exchange", "vc_sharded_add_codes",
    "vc_sharded_add_synthetic", "vc_sharded_size", "vc_sharded_build_index", "vc_sharded_get_code", "vc_sharded_get_bucket",
    "vc_sharded_search_knn", "vc_sharded_shard", "vc_sharded_search_knn_dev", "vc_sharded_root_device", "vc_search_knn_dev_stats", "vc_sharded_search_radius",
]
MAX_SHARDS = 16
EXCHANGE_AUTO, EXCHANGE_PEER_COPY, EXCHANGE_RCCL = 0, 1, 2


class VcConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("bits", C.c_uint32), ("n_tables", C.c_uint32), ("flags", C.c_uint32),
        ("capacity", C.c_uint64), ("id_base", C.c_uint32), ("device", C.c_int32), ("cand_cap", C.c_uint32),
        ("scan_blocks", C.c_uint32), ("query_tile", C.c_uint32), ("timing_sample", C.c_uint32), ("reserved", C.c_uint32 * 4),
    ]


class VcShardedConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("n_shards", C.c_uint32), ("n_devices", C.c_uint32), ("exchange", C.c_uint32),
        ("device_ids", C.c_int32 * 16), ("engine", VcConfig),
    ]


class VcQueryStats(C.Structure):
    _fields_ = [
        ("radius", C.c_uint32), ("n_results", C.c_uint32), ("n_main_reads", C.c_uint64), ("n_sub_reads", C.c_uint64),
        ("n_local_reads", C.c_uint64), ("n_candidates", C.c_uint64),
    ]


class VcTiming(C.Structure):
    _fields_ = [
        ("total_ms", C.c_float), ("scan_ms", C.c_float), ("scan_launches", C.c_uint32), ("calls", C.c_uint32),
        ("scan_bytes", C.c_uint64),
        ("mih_ms", C.c_float), ("mih_launches", C.c_uint32), ("mih_queries", C.c_uint64), ("mih_probes", C.c_uint64),
        ("mih_hits", C.c_uint64), ("mih_entries", C.c_uint64),
    ]


class VcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("verticut_gpu error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """dlopen the in-tree HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m verticut_amd.build` (hipcc, gfx950). "
            "verticut_amd has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so (soname libamdhip64.so.7) and
    # asks for it as "libamdhip64.so"; if our library pulled in /opt/rocm's copy first, torch would load a second
    # runtime next to it and find no GPU.  Loading torch first makes our NEEDED libamdhip64.so.7 resolve to that copy.
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.vc_create.argtypes = [C.POINTER(VcConfig), C.POINTER(vp)]
    L.vc_destroy.argtypes = [vp]
    L.vc_last_error.restype = C.c_char_p
    L.vc_last_error.argtypes = [vp]
    L.vc_strerror.restype = C.c_char_p
    L.vc_strerror.argtypes = [C.c_int]
    L.vc_add_codes.argtypes = [vp, vp, u64]
    L.vc_add_synthetic.argtypes = [vp, u64, u64, u32, u32, u32]
    L.vc_size.argtypes = [vp, C.POINTER(u64)]
    L.vc_get_code.argtypes = [vp, u32, vp]
    L.vc_build_index.argtypes = [vp]
    L.vc_get_bucket.argtypes = [vp, u32, u32, vp, vp, u32, C.POINTER(u32)]
    L.vc_bitmap_test.argtypes = [vp, u32, u32, C.POINTER(C.c_int)]
    L.vc_bitmap_read.argtypes = [vp, u32, u64, u64, vp]
    L.vc_search_knn.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp]
    L.vc_search_knn_dev.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp]
    L.vc_search_knn_dev_stats.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp, vp]
    L.vc_search_radius.argtypes = [vp, vp, u32, u32, u32, vp, u64, vp]
    L.vc_search_radius_dev.argtypes = [vp, vp, u32, u32, u32, vp, u64, vp, vp]
    L.vc_merge_topk_dev.argtypes = [vp, u32, u32, u32, vp, vp, vp]
    L.vc_load_code_file.argtypes = [vp, C.c_char_p, u64, C.POINTER(u64)]
    L.vc_save_code_file.argtypes = [vp, C.c_char_p]
    L.vc_write_bitmap_file.argtypes = [vp, u32, C.c_char_p]
    L.vc_read_bitmap_file.argtypes = [vp, u32, C.c_char_p, C.POINTER(u64)]
    L.vc_save_index.argtypes = [vp, C.c_char_p]
    L.vc_load_index.argtypes = [vp, C.c_char_p]
    L.vc_get_timing.argtypes = [vp, C.POINTER(VcTiming)]
    L.vc_set_stream.argtypes = [vp, vp]
    L.vc_device_status.argtypes = [vp, C.POINTER(u32)]
    L.vc_sharded_create.argtypes = [C.POINTER(VcShardedConfig), C.POINTER(vp)]
    L.vc_sharded_destroy.argtypes = [vp]
    L.vc_sharded_last_error.restype = C.c_char_p
    L.vc_sharded_last_error.argtypes = [vp]
    L.vc_sharded_exchange.argtypes = [vp, C.POINTER(u32)]
    L.vc_sharded_add_codes.argtypes = [vp, vp, u64]
    L.vc_sharded_add_synthetic.argtypes = [vp, u64, u64, u32, u32, u32]
    L.vc_sharded_size.argtypes = [vp, C.POINTER(u64)]
    L.vc_sharded_build_index.argtypes = [vp]
    L.vc_sharded_get_code.argtypes = [vp, u32, vp]
    L.vc_sharded_get_bucket.argtypes = [vp, u32, u32, vp, vp, u32, C.POINTER(u32)]
    L.vc_sharded_search_knn.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp]
    L.vc_sharded_shard.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.vc_sharded_search_knn_dev.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp, vp]
    L.vc_sharded_root_device.argtypes = [vp, C.POINTER(C.c_int)]
    L.vc_sharded_search_radius.argtypes = [vp, vp, u32, u32, u32, vp, u64, vp]
    for name in EXPORTS:
        if getattr(L, name).restype is not C.c_char_p:
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def split(packed):
    """packed uint64 -> (ids uint32, dists uint32)  (GET_ID / GET_DIST, search_worker.cc:12-13)."""
    packed = np.asarray(packed, dtype=np.uint64)
    return (packed & np.uint64(0xFFFFFFFF)).astype(np.uint32), (packed >> np.uint64(32)).astype(np.uint32)


class Engine:
    """One HBM-resident shard of the code database plus its MIH index (one per GPU/process)."""

    def __init__(self, bits, capacity, n_tables=0, flags=0, id_base=0, device=-1, cand_cap=0, scan_blocks=0,
                 query_tile=0, timing_sample=0):
        self._L = load_library()
        self.bits, self.nbytes, self.n_tables, self.id_base = bits, bits // 8, n_tables, id_base
        cfg = VcConfig(abi_version=VC_ABI_VERSION, bits=bits, n_tables=n_tables, flags=flags, capacity=capacity,
                       id_base=id_base, device=device, cand_cap=cand_cap, scan_blocks=scan_blocks,
                       query_tile=query_tile, timing_sample=timing_sample)
        h = C.c_void_p()
        rc = self._L.vc_create(C.byref(cfg), C.byref(h))
        if rc != VC_OK:
            raise VcError(rc, self._L.vc_last_error(None).decode())
        self._h = h

    # -- plumbing
    def _check(self, rc, ok=(VC_OK,)):
        if rc not in ok:
            raise VcError(rc, self._L.vc_last_error(self._h).decode() or self._L.vc_strerror(rc).decode())
        return rc

    @classmethod
    def borrowed(cls, handle, bits, n_tables=0, id_base=0):
        """a view of an engine somebody else owns (a shard of a ShardedEngine): every call works, close() does not destroy"""
        e = cls.__new__(cls)
        e._L = load_library()
        e.bits, e.nbytes, e.n_tables, e.id_base = bits, bits // 8, n_tables, id_base
        e._h, e._borrowed = handle, True
        return e

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._L.vc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _queries(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.nbytes:
            raise ValueError("query must be %d bytes" % self.nbytes)
        return q

    # -- ingest
    def add_codes(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        assert codes.ndim == 2 and codes.shape[1] == self.nbytes
        self._check(self._L.vc_add_codes(self._h, _p(codes), codes.shape[0]))

    def add_synthetic(self, n, seed, kind=SYNTH_UNIFORM, n_centres=0, max_flips=0):
        self._check(self._L.vc_add_synthetic(self._h, n, seed, kind, n_centres, max_flips))

    def load_code_file(self, path, max_records=0):
        """headerless records of bits/8 bytes (the reference's BINARY_CODE_FILE); returns records appended"""
        n = C.c_uint64()
        self._check(self._L.vc_load_code_file(self._h, os.fsencode(path), max_records, C.byref(n)))
        return n.value

    def save_code_file(self, path):
        self._check(self._L.vc_save_code_file(self._h, os.fsencode(path)))

    def write_bitmap_file(self, table, path):
        """raw LSB-first uint32 words of one table's occupancy bitmap (generate_bitmap.cc:122-125 format)"""
        self._check(self._L.vc_write_bitmap_file(self._h, table, os.fsencode(path)))

    def read_bitmap_file(self, table, path):
        """checks a generate_bitmap.cc-format file against the index's bitmap; returns the number of differing words
        (0 = the file belongs to this database); raises for unreadable / wrongly sized files"""
        bad = C.c_uint64()
        self._check(self._L.vc_read_bitmap_file(self._h, table, os.fsencode(path), C.byref(bad)), ok=(VC_OK, VC_ERR_STATE))
        return bad.value

    def save_index(self, path):
        self._check(self._L.vc_save_index(self._h, os.fsencode(path)))

    def load_index(self, path):
        self._check(self._L.vc_load_index(self._h, os.fsencode(path)))

    def __len__(self):
        n = C.c_uint64()
        self._check(self._L.vc_size(self._h, C.byref(n)))
        return n.value

    def get_code(self, gid):
        out = np.empty(self.nbytes, dtype=np.uint8)
        rc = self._check(self._L.vc_get_code(self._h, gid, _p(out)), ok=(VC_OK, VC_NOT_FOUND))
        return out if rc == VC_OK else None

    # -- index
    def build_index(self):
        self._check(self._L.vc_build_index(self._h))

    def get_bucket(self, table, index, cap=1 << 16, with_codes=True):
        """BaseProxy.get(HashIndex{table,index}) -> (ids, codes) or None (PROXY_NOT_FOUND)."""
        ids = np.empty(cap, dtype=np.uint32)
        codes = np.empty((cap, self.nbytes), dtype=np.uint8) if with_codes else None
        n = C.c_uint32()
        rc = self._check(self._L.vc_get_bucket(self._h, table, index, _p(ids), _p(codes) if with_codes else None, cap,
                                               C.byref(n)), ok=(VC_OK, VC_NOT_FOUND))
        if rc == VC_NOT_FOUND:
            return None
        m = min(n.value, cap)
        return ids[:m].copy(), (codes[:m].copy() if with_codes else None), n.value

    def bitmap_test(self, table, index):
        b = C.c_int()
        self._check(self._L.vc_bitmap_test(self._h, table, index, C.byref(b)))
        return b.value

    def bitmap_read(self, table, word_off, n_words):
        out = np.empty(n_words, dtype=np.uint32)
        self._check(self._L.vc_bitmap_read(self._h, table, word_off, n_words, _p(out)))
        return out

    # -- search
    def search_knn(self, queries, k, mode=MODE_LINEAR, order=ORDER_ASCENDING, with_stats=False, out=None, counts=None):
        """Returns (packed [nq,k] uint64, counts [nq]) (+ list of VcQueryStats).
        out / counts: optional preallocated C-contiguous arrays ([nq,k] uint64, [nq] uint32) to receive the results -- e.g. views
        of page-locked memory (torch.empty(...).pin_memory().numpy()): a large batch then leaves the device in one DMA instead of
        through the library's staging chunks (13 MB of rows per 16 384-query top-100 call)."""
        q = self._queries(queries)
        nq = q.shape[0]
        if out is None:
            out = np.empty((nq, k), dtype=np.uint64)     # (the call writes every row in full, PACK_INF behind counts[i] entries)
        elif out.dtype != np.uint64 or out.shape != (nq, k) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous [nq, k] uint64 array")
        if counts is None:
            counts = np.zeros(nq, dtype=np.uint32)
        elif counts.dtype != np.uint32 or counts.shape != (nq,) or not counts.flags.c_contiguous:
            raise ValueError("counts must be a C-contiguous [nq] uint32 array")
        stats = (VcQueryStats * nq)() if with_stats else None
        self._check(self._L.vc_search_knn(self._h, _p(q), nq, k, mode, order, _p(out), _p(counts),
                                          C.cast(stats, C.c_void_p) if with_stats else None))
        if with_stats:
            return out, counts, list(stats)
        return out, counts

    def search_knn_dev(self, d_queries, nq, k, d_out, d_counts=None, mode=MODE_LINEAR, stream=None):
        """Device-pointer variant: arguments are raw device addresses (ints), e.g. tensor.data_ptr().
        stream: raw hipStream_t (torch.cuda.current_stream().cuda_stream); None/0 = the null stream, which is
        PyTorch's default stream, so the call is ordered with torch work either way."""
        self._check(self._L.vc_search_knn_dev(self._h, d_queries, nq, k, mode, d_out, d_counts, stream))

    def search_knn_dev_stats(self, d_queries, nq, k, d_out, d_counts, d_stats, mode=MODE_LINEAR, stream=None):
        """vc_search_knn_dev_stats: the same + nq VcQueryStats records written to device memory (d_stats, raw address) in stream order"""
        self._check(self._L.vc_search_knn_dev_stats(self._h, d_queries, nq, k, mode, d_out, d_counts, d_stats, stream))

    def search_radius(self, queries, radius, mode=MODE_LINEAR, cap_per_query=4096):
        q = self._queries(queries)
        nq = q.shape[0]
        offs = np.zeros(nq + 1, dtype=np.uint64)
        cap = nq * cap_per_query
        for _ in range(2):
            out = np.empty(max(cap, 1), dtype=np.uint64)
            rc = self._check(self._L.vc_search_radius(self._h, _p(q), nq, radius, mode, _p(out), cap, _p(offs)),
                             ok=(VC_OK, VC_ERR_CAPACITY))
            if rc == VC_OK:
                return [out[int(offs[i]):int(offs[i + 1])].copy() for i in range(nq)]
            cap = int(offs[nq])
        raise VcError(VC_ERR_CAPACITY, "radius search output does not fit")

    def search_radius_dev(self, d_queries, nq, radius, d_out, out_cap, d_offsets, mode=MODE_MIH_EXACT, stream=None):
        """Device-pointer radius search (raw device addresses); returns VC_OK or VC_ERR_CAPACITY (d_offsets[nq] = needed)."""
        return self._check(self._L.vc_search_radius_dev(self._h, d_queries, nq, radius, mode, d_out, out_cap, d_offsets, stream),
                           ok=(VC_OK, VC_ERR_CAPACITY))

    def timing(self):
        t = VcTiming()
        self._check(self._L.vc_get_timing(self._h, C.byref(t)))
        return t

    def device_status(self):
        """calls since the last query whose device-side ring-overflow recovery gave up (0 in normal operation)"""
        n = C.c_uint32()
        self._check(self._L.vc_device_status(self._h, C.byref(n)))
        return n.value

    def set_stream(self, stream):
        self._check(self._L.vc_set_stream(self._h, stream))


def merge_topk_dev(d_lists, n_lists, nq, k, d_out, d_counts=None, stream=None):
    """vc_merge_topk_dev on raw device addresses (replaces gather_vectors + master heap)."""
    L = load_library()
    rc = L.vc_merge_topk_dev(d_lists, n_lists, nq, k, d_out, d_counts, stream)
    if rc != VC_OK:
        raise VcError(rc, L.vc_last_error(None).decode())


class ShardedEngine:
    """The vc_sharded_* family: one process, the database split by id range over `n_shards` engines on `devices`
    (replaces mpirun ranks + MPI gathers + the master heap, search_worker.cc:99-101,177-207)."""

    def __init__(self, bits, capacity, n_shards, n_tables=0, devices=None, exchange=EXCHANGE_AUTO, flags=0, id_base=0,
                 cand_cap=0, query_tile=0, timing_sample=0):
        self._L = load_library()
        self.bits, self.nbytes, self.n_shards, self.n_tables = bits, bits // 8, n_shards, n_tables
        cfg = VcShardedConfig(abi_version=VC_ABI_VERSION, n_shards=n_shards, n_devices=len(devices or []), exchange=exchange)
        for i, d in enumerate(devices or []):
            cfg.device_ids[i] = d
        cfg.engine = VcConfig(abi_version=VC_ABI_VERSION, bits=bits, n_tables=n_tables, flags=flags, capacity=capacity,
                              id_base=id_base, device=-1, cand_cap=cand_cap, query_tile=query_tile, timing_sample=timing_sample)
        h = C.c_void_p()
        rc = self._L.vc_sharded_create(C.byref(cfg), C.byref(h))
        if rc != VC_OK:
            raise VcError(rc, self._L.vc_sharded_last_error(None).decode())
        self._h = h

    def _check(self, rc, ok=(VC_OK,)):
        if rc not in ok:
            raise VcError(rc, self._L.vc_sharded_last_error(self._h).decode() or self._L.vc_strerror(rc).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._L.vc_sharded_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def exchange(self):
        k = C.c_uint32()
        self._check(self._L.vc_sharded_exchange(self._h, C.byref(k)))
        return k.value

    def add_codes(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self._check(self._L.vc_sharded_add_codes(self._h, _p(codes), codes.shape[0]))

    def add_synthetic(self, n, seed, kind=SYNTH_UNIFORM, n_centres=0, max_flips=0):
        self._check(self._L.vc_sharded_add_synthetic(self._h, n, seed, kind, n_centres, max_flips))

    def __len__(self):
        n = C.c_uint64()
        self._check(self._L.vc_sharded_size(self._h, C.byref(n)))
        return n.value

    def build_index(self):
        self._check(self._L.vc_sharded_build_index(self._h))

    def get_code(self, gid):
        out = np.empty(self.nbytes, dtype=np.uint8)
        rc = self._check(self._L.vc_sharded_get_code(self._h, gid, _p(out)), ok=(VC_OK, VC_NOT_FOUND))
        return out if rc == VC_OK else None

    def get_bucket(self, table, index, cap=1 << 16):
        ids = np.empty(cap, dtype=np.uint32)
        codes = np.empty((cap, self.nbytes), dtype=np.uint8)
        n = C.c_uint32()
        rc = self._check(self._L.vc_sharded_get_bucket(self._h, table, index, _p(ids), _p(codes), cap, C.byref(n)),
                         ok=(VC_OK, VC_NOT_FOUND))
        if rc == VC_NOT_FOUND:
            return None
        m = min(n.value, cap)
        return ids[:m].copy(), codes[:m].copy(), n.value

    def shard_range(self, g):
        first, cnt = C.c_uint64(), C.c_uint64()
        self._check(self._L.vc_sharded_shard(self._h, g, None, C.byref(first), C.byref(cnt)))
        return first.value, cnt.value

    def shard(self, g):
        """shard g's engine, borrowed (timing, bucket views, files); valid while this store lives"""
        e, first = C.c_void_p(), C.c_uint64()
        self._check(self._L.vc_sharded_shard(self._h, g, C.byref(e), C.byref(first), None))
        return Engine.borrowed(e, self.bits, self.n_tables, first.value)

    @property
    def root_device(self):
        d = C.c_int()
        self._check(self._L.vc_sharded_root_device(self._h, C.byref(d)))
        return d.value

    def search_knn_dev(self, d_queries, nq, k, d_out, d_counts=None, d_stats=None, mode=MODE_LINEAR, stream=None):
        """vc_sharded_search_knn_dev: raw device addresses on the root device; results valid in `stream` order"""
        self._check(self._L.vc_sharded_search_knn_dev(self._h, d_queries, nq, k, mode, d_out, d_counts, d_stats, stream))

    def search_radius(self, queries, radius, mode=MODE_LINEAR, cap_per_query=64):
        """vc_sharded_search_radius: list of ascending packed arrays, one per query"""
        q = np.ascontiguousarray(queries, dtype=np.uint8)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        offs = np.zeros(nq + 1, dtype=np.uint64)
        cap = nq * cap_per_query
        for _ in range(2):
            out = np.empty(max(cap, 1), dtype=np.uint64)
            rc = self._check(self._L.vc_sharded_search_radius(self._h, _p(q), nq, radius, mode, _p(out), cap, _p(offs)),
                             ok=(VC_OK, VC_ERR_CAPACITY))
            if rc == VC_OK:
                return [out[int(offs[i]):int(offs[i + 1])].copy() for i in range(nq)]
            cap = int(offs[nq])
        raise VcError(VC_ERR_CAPACITY, "radius search output does not fit")

    def search_knn(self, queries, k, mode=MODE_LINEAR, order=ORDER_ASCENDING, with_stats=False):
        q = np.ascontiguousarray(queries, dtype=np.uint8)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        out = np.full((nq, k), PACK_INF, dtype=np.uint64)
        counts = np.zeros(nq, dtype=np.uint32)
        stats = (VcQueryStats * nq)() if with_stats else None
        self._check(self._L.vc_sharded_search_knn(self._h, _p(q), nq, k, mode, order, _p(out), _p(counts),
                                                  C.cast(stats, C.c_void_p) if with_stats else None))
        if with_stats:
            return out, counts, list(stats)
        return out, counts
