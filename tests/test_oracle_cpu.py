"""CPU suite: the oracle against the reference-generated golden vectors (and the live reference veneer when
it is built), the oracle's loops against each other, and the C-ABI library's exported surface.
No GPU compute is attempted here."""
import ctypes
import json
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SH = np.uint64(32)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "primitives.json")) as f:
        return json.load(f)


# ----------------------------------------------------------------------------- primitives (pinned)
def test_hamming_golden(oracle, golden):
    for c in golden["hamming"]:
        a = np.frombuffer(bytes.fromhex(c["a"]), dtype=np.uint8)
        b = np.frombuffer(bytes.fromhex(c["b"]), dtype=np.uint8)
        assert oracle.hamming(a, b) == c["dist"]


def test_binary_to_int_golden(oracle, golden):
    for c in golden["binary_to_int"]:
        raw = bytes.fromhex(c["bytes"])
        assert oracle.binary_to_int(raw) == c["value"]
    # the quirk the survey documents (Pilaf/image_tools.h:13): short substrings sign-extend
    assert oracle.binary_to_int(bytes([0x01, 0x80])) == 0xFFFF8001
    assert oracle.binary_to_int(bytes([0x01, 0x7F])) == 0x00007F01


def test_bitmap_golden(oracle, golden):
    L = oracle.lib()
    for c in golden["bitmap"]:
        words = np.zeros(max(c["n_bytes"] // 4, 1), dtype=np.uint32)
        gets = []
        for kind, bit in c["ops"]:
            if kind == "set":
                L.vco_bitmap_set(words.ctypes.data, bit)
            elif kind == "reset":
                L.vco_bitmap_reset(words.ctypes.data, bit)
            else:
                gets.append(L.vco_bitmap_get(words.ctypes.data, bit))
        assert gets == c["gets"]
        assert words.tobytes()[: c["n_bytes"]].hex() == c["raw"]


def test_primitives_against_live_reference(oracle):
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(1)
    for nb in (8, 16, 32, 64, 13):
        for _ in range(300):
            a = rng.integers(0, 256, nb, dtype=np.uint8)
            b = rng.integers(0, 256, nb, dtype=np.uint8)
            assert oracle.hamming(a, b) == R.vcref_hamming(a.tobytes(), b.tobytes(), nb)
    for ln in (1, 2, 3, 4):
        for _ in range(500):
            p = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
            assert oracle.binary_to_int(p) == R.vcref_binary_to_int(p, ln)


# ----------------------------------------------------------------------------- loops (restated; cross-validated)
def test_generator_is_counter_based(oracle):
    a = oracle.gen_codes(1000, 128, 34)
    b = oracle.gen_codes(300, 128, 34, first_id=500)
    assert np.array_equal(a[500:800], b)
    c = oracle.gen_codes(1000, 256, 34, kind=1, n_centres=10, max_flips=5)
    d = oracle.gen_codes(10, 256, 34, kind=1, n_centres=10, max_flips=5, first_id=990)
    assert np.array_equal(c[990:], d)
    assert len({bytes(r) for r in c}) > 10  # flips applied


@pytest.mark.parametrize("bits", [64, 128, 256])
def test_linear_reference_order_vs_canonical(oracle, bits):
    rng = np.random.default_rng(bits)
    codes = oracle.gen_codes(20000, bits, 34, kind=1, n_centres=50, max_flips=8)
    for k in (1, 10, 100):
        q = codes[rng.integers(0, 20000)].copy()
        q[0] ^= 0x81
        ref = oracle.linear_knn_ref(codes, q, k)       # linear_search.cc:59-63: farthest first
        can = oracle.linear_knn(codes, q, k)
        mt = oracle.linear_knn(codes, q, k, threads=3)
        assert np.array_equal(can, mt)
        d = oracle.np_distances(codes, q)
        brute = np.sort(oracle.pack(d, np.arange(20000, dtype=np.uint64)))[:k]
        assert np.array_equal(can, brute)
        assert np.all(np.diff((ref >> SH).astype(np.int64)) <= 0)
        assert np.array_equal(np.sort(ref >> SH), can >> SH)
        dk = can[-1] >> SH
        assert set(ref[(ref >> SH) < dk].tolist()) == set(can[(can >> SH) < dk].tolist())


def test_fewer_items_than_k(oracle):
    # 64-bit codes / 16-bit substrings: seven uniform items sit ~32 bits apart, so the radius loop walks to shell ~7 of
    # 65 536-key tables (with 32-bit substrings the same test enumerated 1.5e9 leaves: 116 s of the CPU suite)
    codes = oracle.gen_codes(7, 64, 1)
    assert len(oracle.linear_knn_ref(codes, codes[0], 10)) == 7
    assert len(oracle.linear_knn(codes, codes[0], 10)) == 7
    mo = oracle.MihOracle(codes, 4, key_mode=1)
    res, st = mo.find(codes[0], 3, stop_mult=4)
    assert len(res) == 3


@pytest.mark.parametrize("bits,m", [(128, 4), (64, 4), (256, 8)])
def test_mih_equals_linear_and_stats_formula(oracle, bits, m):
    """exact MIH returns the linear scan's distance multiset (what the survey observed on the compiled
    reference), and without a bitmap n_sub_reads = sum_{r' <= radius} C(s, r') (1, 33, 529, ... for s = 32)."""
    s = bits // m
    n = 30000
    rng = np.random.default_rng(m)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=150, max_flips=2 * m)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    for k in (1, 10, 100):
        q = codes[rng.integers(0, n)].copy()
        q[1] ^= 0x3
        res, st = mo.find(q, k, stop_mult=4)
        lin = oracle.linear_knn(codes, q, k)
        assert np.array_equal(np.sort(res >> SH), lin >> SH)
        dk = lin[-1] >> SH
        assert set(res[(res >> SH) < dk].tolist()) == set(lin[(lin >> SH) < dk].tolist())
        assert st.n_sub_reads == sum(math.comb(s, r) for r in range(st.radius + 1))
        assert st.n_local_reads == 0 and st.n_main_reads == 0
        # stop rule (search_worker.cc:201-205): k-th distance <= 4 * (radius + 1), and not yet at radius - 1
        assert int(np.max(res >> SH)) <= 4 * (st.radius + 1)
        res_b, st_b = mo.find(q, k, use_bitmap=True, stop_mult=4)
        assert st_b.radius == st.radius and st_b.n_local_reads == st.n_sub_reads and st_b.n_sub_reads <= st.n_sub_reads


def test_mih_approximate_restatement(oracle):
    """order-independent reading of search_worker.cc:93-157: stop after the first shell at which >= 20k distinct
    items have been seen; answer = the k smallest distances among everything seen."""
    n, bits, m, k = 40000, 128, 4, 10
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=60, max_flips=10)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    rng = np.random.default_rng(4)
    for _ in range(4):
        q = codes[rng.integers(0, n)].copy()
        res, st = mo.find(q, k, approximate=True)
        mind = oracle.np_sub_distances(codes, q, m).min(axis=1)
        d = oracle.np_distances(codes, q)
        r_stop = next(r for r in range(33) if (mind <= r).sum() >= 20 * k or r == 32)
        assert st.radius == r_stop
        assert st.n_distinct == int((mind <= r_stop).sum())
        assert np.array_equal(np.sort(res >> SH), np.sort(d[mind <= r_stop])[:k])


def test_signext_quirk_changes_reachability(oracle):
    """16-bit substrings: with reference keys a probe that flips bit 15 can never match, so an item whose
    substrings all differ from the query in their top bit is only reachable with masked keys."""
    base = np.zeros(8, dtype=np.uint8)
    other = base.copy()
    other[[1, 3, 5, 7]] = 0x80           # top bit of each of the four 16-bit substrings
    codes = np.stack([other, other])
    ref_keys = oracle.MihOracle(codes, 4, key_mode=0)
    masked = oracle.MihOracle(codes, 4, key_mode=1)
    r0, s0 = ref_keys.find(base, 1, stop_mult=4)
    r1, s1 = masked.find(base, 1, stop_mult=4)
    assert len(r1) == 1 and int(r1[0] >> SH) == 4 and s1.radius == 1
    assert len(r0) == 0 and s0.radius == 16   # never found: loop runs to the last shell


# ----------------------------------------------------------------------------- C-ABI surface (no GPU needed)
def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "verticut_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vc_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(vc):
    declared = _declared_symbols()
    assert declared == sorted(vc.EXPORTS)
    L = ctypes.CDLL(vc.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert vc.load_library().vc_abi_version() == vc.VC_ABI_VERSION
    assert vc.load_library().vc_strerror(vc.VC_ERR_NO_DEVICE).decode().startswith("no usable")


def test_config_struct_layout_matches_header(vc):
    assert ctypes.sizeof(vc.VcConfig) == 64 and ctypes.sizeof(vc.VcQueryStats) == 40 and ctypes.sizeof(vc.VcTiming) == 64
    assert ctypes.sizeof(vc.VcShardedConfig) == 16 + 16 * 4 + 64 and vc.VcShardedConfig.engine.offset == 80


def test_product_never_imports_the_oracle():
    """the oracle is the checker, never the thing shipped: nothing under verticut_amd/, include/ or tools/ names it
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may)."""
    bad = []
    for base in ("verticut_amd", "include", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".cc")):
                    src = open(os.path.join(dp, fn), errors="replace").read()
                    if re.search(r"^\s*(from|import)\s+(oracle|vc_oracle)|libvcoracle|libvcref|dlopen\(.*oracle", src, flags=re.M):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


# ----------------------------------------------------------------------------- committed search fixtures
def test_oracle_reproduces_search_fixtures(oracle):
    """tests/golden/search_fixtures.json (made by make_search_fixtures.py from the oracle) still holds: guards the
    generator definition and the loops against drift between the authoring container and any other box."""
    with open(os.path.join(ROOT, "tests", "golden", "search_fixtures.json")) as f:
        fxs = json.load(f)["fixtures"]
    for fx in fxs[:3]:   # the GPU suite covers all of them; keep the CPU suite short
        codes = oracle.gen_codes(fx["n"], fx["bits"], fx["seed"], fx["kind"], fx["n_centres"], fx["max_flips"])
        for i, qh in enumerate(fx["queries"]):
            q = np.frombuffer(bytes.fromhex(qh), dtype=np.uint8)
            assert [int(v) for v in oracle.linear_knn(codes, q, fx["k"])] == fx["linear"][i]
        if fx["mih_exact"]:
            mo = oracle.MihOracle(codes, fx["m"], key_mode=1)
            for i, qh in enumerate(fx["queries"]):
                q = np.frombuffer(bytes.fromhex(qh), dtype=np.uint8)
                res, st = mo.find(q, fx["k"], stop_mult=min(fx["m"], 4))
                e = fx["mih_exact"][i]
                assert (st.radius, st.n_sub_reads, st.n_distinct) == (e["radius"], e["n_sub_reads"], e["n_candidates"])
                assert sorted(int(v) >> 32 for v in res) == [v >> 32 for v in e["result"]]


# ----------------------------------------------------------------------------- protobuf wire records (host layer)
def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_wire_encoding_matches_proto2(tmp_path):
    """verticut_wire.hpp writes the bytes protobuf would for image_search.proto:3-27.
    Pinned two ways: tests/golden/wire_vectors.json holds field values and the bytes a REAL protobuf runtime serialized
    them to (tests/golden/make_wire_vectors.py, google.protobuf in the authoring container) -- every vector must come out
    of vc::wire::encode byte for byte and survive decode -> encode; and a few strings assembled here from the proto2
    wire rules (kept from round 1: an independent second reading)."""
    import subprocess
    exe = tmp_path / "wire_test"
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "wire_test.cc"),
                           "-I", os.path.join(ROOT, "verticut_amd", "host")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = dict(l.split(" ", 1) for l in out.strip().splitlines())
    code = b"0123456789123456"
    pair = lambda i: b"\x08" + _varint(i) + b"\x12" + _varint(len(code)) + code   # noqa: E731
    assert got["id"] == (b"\x08" + _varint(300)).hex()
    assert got["hashindex"] == (b"\x08" + _varint(3) + b"\x10" + _varint(0xFFFF8001)).hex()
    assert got["binarycode"] == (b"\x0a" + _varint(16) + code).hex()
    assert got["imagelist"] == b"".join(b"\x0a" + _varint(len(pair(i))) + pair(i) for i in (0, 1000000, 2000000)).hex()
    assert got["roundtrip"] == "ok"

    with open(os.path.join(ROOT, "tests", "golden", "wire_vectors.json")) as f:
        vec = json.load(f)
    lines, want = [], []
    for v in vec["id"]:
        lines.append("id %d" % v["id"])
        want.append(v["wire"])
    for v in vec["binarycode"]:
        lines.append("binarycode %s" % (v["code"] or "-"))
        want.append(v["wire"])
    for v in vec["hashindex"]:
        lines.append("hashindex %d %d" % (v["table_id"], v["index"]))
        want.append(v["wire"])
    for v in vec["imagelist"]:
        lines.append("imagelist %d %s" % (len(v["images"]), " ".join("%d %s" % (i, c or "-") for i, c in v["images"])))
        want.append(v["wire"])
    assert len(want) >= 60
    res = subprocess.run([str(exe), "--vectors"], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in res.strip().splitlines()]
    assert len(rows) == len(want)
    for line, (enc, again), w in zip(lines, rows, want):
        assert enc == (w or "-"), "%s: verticut_wire.hpp wrote %s, protobuf wrote %s" % (line, enc, w)
        assert again == enc, "%s: decode -> encode changed the bytes" % line


def test_oracle_threaded_find_and_radius_search(oracle):
    """round-2 oracle entry points (bench.py's CPU-MIH baselines): the m ranks of a radius iteration on m threads give
    exactly the single-thread result (rank-order gather, mpi_coordinator.cc:34-69), and the fixed-radius neighbour
    search (search_R_neighbors shells 0..r/m + gather + dedup, search_worker.cc:222-264) equals numpy brute force."""
    import numpy as np
    codes = oracle.gen_codes(30000, 64, 34, kind=1, n_centres=80, max_flips=6)
    rng = np.random.default_rng(3)
    for m, key_mode in ((2, 1), (4, 1), (4, 0)):
        mo = oracle.MihOracle(codes, m, key_mode=key_mode)
        for qi in rng.integers(0, len(codes), size=5):
            q = codes[qi].copy()
            q[int(rng.integers(0, 8))] ^= np.uint8(1 << int(rng.integers(0, 8)))
            a, sa = mo.find(q, 10, stop_mult=min(m, 4))
            b, sb = mo.find(q, 10, stop_mult=min(m, 4), threads=m)
            assert np.array_equal(a, b)
            assert (sa.radius, sa.n_sub_reads, sa.n_sub_reads_all, sa.n_distinct) == (sb.radius, sb.n_sub_reads, sb.n_sub_reads_all, sb.n_distinct)
            if key_mode == 1:                       # masked keys: MIH radius search is exact
                for threads in (1, m):
                    r, probes = mo.radius(q, 8, threads=threads)
                    d = oracle.np_distances(codes, q)
                    ids = np.nonzero(d <= 8)[0]
                    assert np.array_equal(r, np.sort(oracle.pack(d[ids], ids.astype(np.uint64))))
                    import math
                    # R = m q + a: tables 0..a search substring radius q, the others q - 1
                    q_, a_ = divmod(8, m)
                    assert probes == sum(math.comb(64 // m, j) for t in range(m) for j in range((q_ if t <= a_ else q_ - 1) + 1))


def test_pool_helpers_equal_the_single_thread_oracle(oracle):
    """the worker-pool legs used at full size (slabbed generation + batched linear_search.cc scan + brute-force radius
    search) are the single-thread oracle, slab boundaries and worker ranges included"""
    n = 200_003
    rng = np.random.default_rng(1)
    with oracle.Pool(5) as P:
        for kw in ({}, {"kind": 1, "n_centres": 50, "max_flips": 9}):
            a = oracle.gen_codes(n, 128, 34, **kw)
            assert np.array_equal(a, P.gen_codes(n, 128, 34, **kw))
            assert np.array_equal(a[1000:5000], P.gen_codes(4000, 128, 34, first_id=1000, **kw))
            q = rng.integers(0, 256, size=(5, 16), dtype=np.uint8)
            q[0] = a[77]
            out, cnt = P.linear_knn(a, q, 100, id_base=7)
            for i in range(5):
                assert np.array_equal(out[i], oracle.linear_knn(a, q[i], 100, id_base=7))
            assert np.array_equal(oracle.linear_knn_slabbed(P, n, 128, 34, q, 100, slab=70_001, **kw),
                                  P.linear_knn(a, q, 100)[0])
        out, cnt = P.linear_knn(a[:30], q, 100)          # fewer records than k: padded rows, true counts
        assert np.all(cnt == 30) and np.all(out[:, 30:] == np.uint64(0xFFFFFFFFFFFFFFFF))
        c64 = oracle.gen_codes(n, 64, 34)
        q64 = c64[[5, 9]].copy()
        q64[0, 0] ^= 3
        for i, r in enumerate(P.linear_radius(c64, q64, 20, cap=8)):   # cap too small at first: the wrapper retries
            d = oracle.np_distances(c64, q64[i])
            hit = np.nonzero(d <= 20)[0]
            assert np.array_equal(r, np.sort(oracle.pack(d[hit], hit.astype(np.uint32))))
        rs = oracle.linear_radius_slabbed(P, n, 64, 34, q64, 20, slab=64_000)
        assert all(np.array_equal(x, y) for x, y in zip(rs, P.linear_radius(c64, q64, 20)))
