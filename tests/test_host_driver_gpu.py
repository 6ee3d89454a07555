"""The C++ host layer (verticut_amd/host: BaseProxy / SearchWorker / image_search_client shapes + the
distributed-image-search driver with the reference's argv) against the oracle, on the GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "verticut_amd", "bin", "distributed-image-search")
SH = np.uint64(32)


def _run(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([DRIVER] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_driver_file_queries_match_oracle(vc, oracle, tmp_path):
    assert os.path.exists(DRIVER), "build() must have produced the host driver"
    n, bits, m, k = 40000, 128, 4, 10
    rng = np.random.default_rng(21)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=200, max_flips=8)
    q = codes[rng.integers(0, n, size=7)].copy()
    q[:, 2] ^= 0x41
    (tmp_path / "lsh.code").write_bytes(codes.tobytes())        # BINARY_CODE_FILE format: headerless records
    (tmp_path / "query.code").write_bytes(q.tobytes())
    out = _run([tmp_path / "lsh.code", n, bits, bits // m, k, "pilaf", 0, 0, -1, tmp_path / "query.code"],
               {"VC_PRINT_RESULTS": "1"})
    blocks = re.split(r"^query \d+\n", out, flags=re.M)[1:]
    assert len(blocks) == len(q)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    rad_sum = sub_sum = 0
    for i, blk in enumerate(blocks):
        pairs = [(int(a), int(b)) for a, b in re.findall(r"^(\d+) : (\d+)$", blk, flags=re.M)]
        st = re.search(r"n_sub_reads : (\d+), n_local_reads : (\d+), radius : (\d+)", blk)
        ores, ost = mo.find(q[i], k, stop_mult=4)
        assert [d for _, d in pairs] == [int(x >> SH) for x in ores]       # farthest first, same distances
        dk = max(d for _, d in pairs)
        assert {a for a, d in pairs if d < dk} == {int(x & np.uint64(0xFFFFFFFF)) for x in ores if int(x >> SH) < dk}
        for a, d in pairs:                                                  # every reported pair is genuine
            assert oracle.hamming(codes[a], q[i]) == d
        assert (int(st.group(1)), int(st.group(2)), int(st.group(3))) == (ost.n_sub_reads, 0, ost.radius)
        rad_sum += ost.radius
        sub_sum += ost.n_sub_reads
    avg = re.search(r"n_sub_reads : (\d+), n_local_reads : (\d+), radius : (\d+), rdma", out.split("Averate result")[1])
    assert (int(avg.group(1)), int(avg.group(3))) == (sub_sum // len(q), rad_sum // len(q))
    # the driver searches the file's queries in ONE batched call (SearchWorker::find_batch); the reference's one-find-per-
    # query sequence prints the same lines
    one = _run([tmp_path / "lsh.code", n, bits, bits // m, k, "pilaf", 0, 0, -1, tmp_path / "query.code"],
               {"VC_PRINT_RESULTS": "1", "VC_QUERY_BY_QUERY": "1"})
    strip = lambda t: [ln for ln in t.splitlines() if not ln.startswith("while :")]
    assert strip(one) == strip(out)


def test_driver_query_by_id(vc, oracle, tmp_path):
    """image_search_client::search_image_by_id: the path the reference ships dead (distributed_image_search.cc:116)."""
    n, bits, m, k = 20000, 128, 4, 5
    codes = oracle.gen_codes(n, bits, 7, kind=1, n_centres=100, max_flips=6)
    (tmp_path / "lsh.code").write_bytes(codes.tobytes())
    out = _run([tmp_path / "lsh.code", n, bits, bits // m, k, "pilaf", 0, 0, 1234])
    pairs = [(int(a), int(b)) for a, b in re.findall(r"^(\d+) : (\d+)$", out, flags=re.M)]
    exp = oracle.linear_knn(codes, codes[1234], k)
    assert sorted(d for _, d in pairs) == [int(x >> SH) for x in exp]
    assert pairs[-1][1] == 0 and (1234, 0) in pairs     # nearest last: the image itself at distance 0


def test_accuracy_test_metrics(vc, oracle, tmp_path):
    """accuracy-test prints the reference's running metrics (accuracy_test.cc:88-91, 106-135)."""
    n, bits, m, k = 30000, 128, 4, 10
    rng = np.random.default_rng(33)
    codes = oracle.gen_codes(n, bits, 11, kind=1, n_centres=60, max_flips=10)
    q = codes[rng.integers(0, n, size=5)].copy()
    q[:, 7] ^= 0x09
    (tmp_path / "lsh.code").write_bytes(codes.tobytes())
    (tmp_path / "query.code").write_bytes(q.tobytes())
    exe = os.path.join(ROOT, "verticut_amd", "bin", "accuracy-test")
    p = subprocess.run([exe, str(tmp_path / "lsh.code"), str(n), str(bits), str(bits // m), str(k), "pilaf", "0", "0", "0",
                        str(tmp_path / "query.code")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if re.match(r"^[\d.]+ [\d.]+ [\d.e-]+$", l)]
    assert len(lines) == len(q)
    mo = oracle.MihOracle(codes, m, key_mode=1)
    tot_ex = tot_app = inacc = 0
    for i in range(len(q)):
        ex, _ = mo.find(q[i], k, stop_mult=4)
        app, _ = mo.find(q[i], k, approximate=True)
        dex, dapp = [int(v >> SH) for v in ex], sorted((int(v >> SH) for v in app), reverse=True)
        tot_ex += sum(dex)
        tot_app += sum(dapp)
        inacc += next((j for j, d in enumerate(dapp) if d <= dex[0]), len(dapp))   # farthest-first lists
        a, b, c = (float(x) for x in lines[i].split())
        assert abs(a - tot_ex / (i + 1) / k) < 1e-3 and abs(b - tot_app / (i + 1) / k) < 1e-3
        assert abs(c - inacc / (i + 1) / k) < 1e-4


def test_code_and_bitmap_file_formats(vc, oracle, tmp_path):
    """the reference's on-disk inputs: headerless code file in, same bytes out; bitmap file = raw LSB-first words."""
    n, bits, m = 20000, 64, 4
    codes = oracle.gen_codes(n, bits, 5, kind=1, n_centres=50, max_flips=4)
    (tmp_path / "lsh.code").write_bytes(codes.tobytes() + b"\x01\x02\x03")     # trailing partial record is ignored
    with vc.Engine(bits, capacity=n + 5000, n_tables=m) as e:
        assert e.load_code_file(tmp_path / "lsh.code", max_records=5000) == 5000
        assert e.load_code_file(tmp_path / "lsh.code") == n                    # reads from the start again: ids continue
        assert len(e) == n + 5000 and np.array_equal(e.get_code(5000), codes[0])
        with pytest.raises(vc.VcError) as ei:                                  # image_total exceeded
            e.load_code_file(tmp_path / "lsh.code", max_records=1)
        assert ei.value.code == vc.VC_ERR_CAPACITY
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        assert e.load_code_file(tmp_path / "lsh.code") == n
        e.save_code_file(tmp_path / "out.code")
        assert (tmp_path / "out.code").read_bytes() == codes.tobytes()
        e.build_index()
        for t in range(m):
            e.write_bitmap_file(t, tmp_path / ("lsh.code_bmp_%d_2b_4k.raw" % (t + 1)))   # generate_bitmap.cc:84-87 names
            words = np.frombuffer((tmp_path / ("lsh.code_bmp_%d_2b_4k.raw" % (t + 1))).read_bytes(), dtype=np.uint32)
            assert words.size == (1 << 16) // 32
            keys = {int.from_bytes(bytes(c[2 * t:2 * t + 2]), "little") for c in codes}
            got = {w * 32 + b for w in np.nonzero(words)[0] for b in range(32) if (int(words[w]) >> b) & 1}
            assert got == keys


def test_gpu_proxy_behaves_like_a_base_proxy(vc, oracle, tmp_path):
    """vc::GpuProxy (verticut_host.hpp) through a C++ caller: put(ID,BinaryCode), get(HashIndex,Image_List),
    get(ID,BinaryCode) with the reference's return codes (base_proxy.h:10-13)."""
    n, bits, m = 3000, 128, 4
    codes = oracle.gen_codes(n, bits, 21, kind=1, n_centres=12, max_flips=2)
    (tmp_path / "lsh.code").write_bytes(codes.tobytes())
    exe = tmp_path / "proxy_dump"
    lib = os.path.join(ROOT, "verticut_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "proxy_dump.cc"),
                           "-I", os.path.join(ROOT, "verticut_amd", "host"), "-L", lib, "-lverticut_gpu",
                           "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib"])
    mo = oracle.MihOracle(codes, m, key_mode=1)
    probes = [(t, mo.key(codes[i], t)) for t, i in ((0, 5), (1, 77), (2, 1234), (3, 2999))] + [(0, 123456789)]
    args = [str(x) for p in probes for x in p]
    p = subprocess.run([str(exe), str(tmp_path / "lsh.code"), str(n), str(bits), str(m)] + args,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = p.stdout
    assert "put_out_of_order 1" in out                      # PROXY_PUT_FAIL
    blocks = re.split(r"^bucket ", out, flags=re.M)[1:]
    for (t, idx), blk in zip(probes, blocks):
        head = blk.splitlines()[0]
        exp = mo.bucket(t, idx)
        if len(exp) == 0:
            assert head == "%d %d rc=1 n=0" % (t, idx)      # PROXY_NOT_FOUND
            continue
        assert head == "%d %d rc=0 n=%d" % (t, idx, len(exp))
        rows = re.findall(r"^  (\d+) ([0-9a-f]+)$", blk, flags=re.M)
        assert [int(a) for a, _ in rows] == list(exp)
        assert all(bytes.fromhex(h) == codes[int(a)].tobytes() for a, h in rows)
    assert ("get_id7 rc=0 " + codes[7].tobytes().hex()) in out
    assert "get_missing rc=1" in out

    # serialized KV view == protobuf bytes of the same records (image_search.proto:11-26)
    def varint(v):
        b = bytearray()
        while v >= 0x80:
            b.append((v & 0x7F) | 0x80)
            v >>= 7
        b.append(v)
        return bytes(b)
    t0, idx0 = probes[0]
    exp_list = b""
    for gid in mo.bucket(t0, idx0):
        pair = b"\x08" + varint(int(gid)) + b"\x12" + varint(16) + codes[int(gid)].tobytes()
        exp_list += b"\x0a" + varint(len(pair)) + pair
    assert ("kv_bucket rc=0 " + exp_list.hex()) in out
    assert ("kv_id7 rc=0 " + (b"\x0a" + varint(16) + codes[7].tobytes()).hex()) in out


@pytest.mark.parametrize("bits,m", [(64, 4), (64, 2)])
def test_driver_reference_quirks_mode(vc, oracle, tmp_path, bits, m):
    """VC_REF_QUIRKS=1: the drivers reproduce the reference where it is NOT exact -- 16-bit substrings with binaryToInt's
    sign-extended keys (Pilaf/image_tools.h:13) and two tables with the stop rule's literal 4 (search_worker.cc:204):
    distances, radius and n_sub_reads equal MihOracle(key_mode=0) / stop_mult=4; the default mode equals the exact
    variant of the oracle instead."""
    n, k = 30000, 10
    rng = np.random.default_rng(5 + m)
    codes = oracle.gen_codes(n, bits, 34, kind=1, n_centres=150, max_flips=6)
    q = codes[rng.integers(0, n, size=9)].copy()
    q[:, 1] ^= 0x90                       # flips the top bit of a 16-bit substring: where the sign extension bites
    (tmp_path / "lsh.code").write_bytes(codes.tobytes())
    (tmp_path / "query.code").write_bytes(q.tobytes())
    for quirks, key_mode, stop_mult in (("1", 0, 4), ("0", 1, min(m, 4))):
        out = _run([tmp_path / "lsh.code", n, bits, bits // m, k, "pilaf", 0, 0, -1, tmp_path / "query.code"],
                   {"VC_PRINT_RESULTS": "1", "VC_REF_QUIRKS": quirks})
        blocks = re.split(r"^query \d+\n", out, flags=re.M)[1:]
        mo = oracle.MihOracle(codes, m, key_mode=key_mode)
        for i, blk in enumerate(blocks):
            pairs = [(int(a), int(b)) for a, b in re.findall(r"^(\d+) : (\d+)$", blk, flags=re.M)]
            st = re.search(r"n_sub_reads : (\d+), n_local_reads : (\d+), radius : (\d+)", blk)
            ores, ost = mo.find(q[i], k, stop_mult=stop_mult)
            assert [d for _, d in pairs] == [int(x >> SH) for x in ores], (quirks, i)
            assert (int(st.group(1)), int(st.group(3))) == (ost.n_sub_reads, ost.radius), (quirks, i)
            for a, d in pairs:
                assert oracle.hamming(codes[a], q[i]) == d


def test_bitmap_file_read_back_and_verify(vc, oracle, tmp_path):
    """generate_bitmap.cc:99-125 writes <code file>_bmp_<t>_2b_4k.raw, bitmap_deamon.cc:41-65 reads it back: a file the engine
    wrote verifies, a HAND-MADE file (bits set one by one through the oracle's ImageBitmap::set_idx restatement, which
    is pinned to the reference's bitmap.cc) verifies, and a file with one foreign bit, or of the wrong size, does not."""
    n, bits, m = 25000, 64, 4
    codes = oracle.gen_codes(n, bits, 11, kind=1, n_centres=60, max_flips=5)
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        e.build_index()
        for t in range(m):
            own = tmp_path / ("own_%d.raw" % t)
            e.write_bitmap_file(t, own)
            assert e.read_bitmap_file(t, own) == 0
            hand = np.zeros((1 << 16) // 32, dtype=np.uint32)
            for c in codes:                                                    # generate_bitmap.cc:111-114: one set_idx per record
                oracle.lib().vco_bitmap_set(hand.ctypes.data, int.from_bytes(bytes(c[2 * t:2 * t + 2]), "little"))
            (tmp_path / "hand.raw").write_bytes(hand.tobytes())
            assert e.read_bitmap_file(t, tmp_path / "hand.raw") == 0
        free_key = next(k for k in range(1 << 16) if not e.bitmap_test(0, k))
        bad = np.frombuffer((tmp_path / "own_0.raw").read_bytes(), dtype=np.uint32).copy()
        oracle.lib().vco_bitmap_set(bad.ctypes.data, free_key)
        (tmp_path / "bad.raw").write_bytes(bad.tobytes())
        assert e.read_bitmap_file(0, tmp_path / "bad.raw") == 1                # one word differs
        assert e.read_bitmap_file(1, tmp_path / "own_0.raw") > 0               # another table's file
        (tmp_path / "short.raw").write_bytes(bad.tobytes()[:-4])
        with pytest.raises(vc.VcError) as ei:
            e.read_bitmap_file(0, tmp_path / "short.raw")
        assert ei.value.code == vc.VC_ERR_INVALID
        (tmp_path / "long.raw").write_bytes(bad.tobytes() + b"\0\0\0\0")
        with pytest.raises(vc.VcError):
            e.read_bitmap_file(0, tmp_path / "long.raw")


@pytest.mark.parametrize("bits,m", [(128, 4), (64, 4)])
def test_index_save_and_load(vc, oracle, tmp_path, bits, m):
    """the built index (bucket lists of build_hash_tables.cc:36-64 as id runs + offsets, bitmaps, rank directories)
    survives a round trip through a file: same buckets, same search results and statistics; a file built for another
    database shape is refused."""
    n, k = 30000, 20
    rng = np.random.default_rng(bits)
    codes = oracle.gen_codes(n, bits, 3, kind=1, n_centres=120, max_flips=6)
    q = codes[rng.integers(0, n, size=8)].copy()
    q[:, 0] ^= 0x06
    path = tmp_path / "index.vcidx"
    with vc.Engine(bits, capacity=n, n_tables=m, id_base=7) as e:
        e.add_codes(codes)
        e.build_index()
        ref = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        rad = e.search_radius(q[:3], 7, mode=vc.MODE_MIH_EXACT)
        e.save_index(path)
    with vc.Engine(bits, capacity=n, n_tables=m, id_base=7) as e2:
        e2.add_codes(codes)
        e2.load_index(path)                                                    # no vc_build_index here
        got = e2.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        assert [(s.radius, s.n_sub_reads, s.n_candidates) for s in got[2]] == [(s.radius, s.n_sub_reads, s.n_candidates) for s in ref[2]]
        assert all(np.array_equal(a, b) for a, b in zip(e2.search_radius(q[:3], 7, mode=vc.MODE_MIH_EXACT), rad))
        mo = oracle.MihOracle(codes, m, key_mode=1, id_base=7)
        for t in range(m):
            key = mo.key(codes[123], t)
            ids, bc, total = e2.get_bucket(t, key)
            assert np.array_equal(ids, mo.bucket(t, key)) and total == len(ids)
    with vc.Engine(bits, capacity=n, n_tables=m, id_base=7) as e3:             # fewer records resident than the file indexes
        e3.add_codes(codes[:-1])
        with pytest.raises(vc.VcError) as ei:
            e3.load_index(path)
        assert ei.value.code == vc.VC_ERR_STATE
    (tmp_path / "junk").write_bytes(b"not an index" * 10)
    with vc.Engine(bits, capacity=n, n_tables=m) as e4:
        e4.add_codes(codes)
        with pytest.raises(vc.VcError) as ei:
            e4.load_index(tmp_path / "junk")
        assert ei.value.code == vc.VC_ERR_INVALID


def test_index_file_is_validated_on_load(vc, oracle, tmp_path):
    """vc_load_index takes nothing from a file unchecked (build_hash_tables.cc's KV rows carried integrity words of their
    own, Pilaf/integrity.h): an index saved from ANOTHER database of the same shape is refused by its code checksum, a
    file with damaged id runs / offsets / bitmap fails the device validation pass, truncated and padded files are
    refused by their size -- and the engine stays usable (build_index afterwards works)."""
    n, bits, m, k = 20000, 128, 4, 10
    codes = oracle.gen_codes(n, bits, 3, kind=1, n_centres=80, max_flips=6)
    other = oracle.gen_codes(n, bits, 4, kind=1, n_centres=80, max_flips=6)
    good, foreign = tmp_path / "good.vcidx", tmp_path / "foreign.vcidx"
    for c, path in ((codes, good), (other, foreign)):
        with vc.Engine(bits, capacity=n, n_tables=m) as e:
            e.add_codes(c)
            e.build_index()
            e.save_index(path)
    raw = bytearray(good.read_bytes())
    header = 56                                            # magic 8 + 6 x u32 + n + checksum + file_bytes
    assert int.from_bytes(raw[48:56], "little") == len(raw)
    ids0 = header + 16                                     # table 0: {n_unique, offsets length}, then ids[n]

    def damaged(name, mutate):
        b = bytearray(raw)
        mutate(b)
        path = tmp_path / name
        path.write_bytes(bytes(b))
        return path

    def swap_ids(b):                                       # still a permutation, but two entries sit in the wrong buckets
        b[ids0:ids0 + 4], b[ids0 + 4 * 9000:ids0 + 4 * 9000 + 4] = b[ids0 + 4 * 9000:ids0 + 4 * 9000 + 4], b[ids0:ids0 + 4]

    def big_id(b):
        b[ids0 + 40:ids0 + 44] = (0xFFFFFFF0).to_bytes(4, "little")

    def dup_id(b):
        b[ids0 + 4:ids0 + 8] = b[ids0:ids0 + 4]

    def bad_offsets(b):
        o = ids0 + 4 * n                                   # offsets[] of table 0
        b[o + 8:o + 12] = (n + 5).to_bytes(4, "little")

    def clear_bitmap_word(b):
        nu = int.from_bytes(raw[header:header + 8], "little")
        bm = ids0 + 4 * n + 4 * (nu + 1)
        i = next(i for i in range(bm, bm + (1 << 29), 4) if raw[i:i + 4] != b"\0\0\0\0")
        b[i:i + 4] = b"\0\0\0\0"

    cases = [(foreign, vc.VC_ERR_STATE), (damaged("swap", swap_ids), vc.VC_ERR_STATE), (damaged("bigid", big_id), vc.VC_ERR_STATE),
             (damaged("dup", dup_id), vc.VC_ERR_STATE), (damaged("offs", bad_offsets), vc.VC_ERR_STATE),
             (damaged("bitmap", clear_bitmap_word), vc.VC_ERR_STATE),
             (damaged("trailing", lambda b: b.extend(b"xx")), vc.VC_ERR_INVALID),
             (damaged("short", lambda b: b.__delitem__(slice(len(b) - 4096, len(b)))), vc.VC_ERR_INVALID)]
    q = codes[[5, 77]].copy()
    q[:, 3] ^= 0x21
    with vc.Engine(bits, capacity=n, n_tables=m) as e:
        e.add_codes(codes)
        for path, code in cases:
            with pytest.raises(vc.VcError) as ei:
                e.load_index(path)
            assert ei.value.code == code, (path.name, ei.value)
        e.load_index(good)                                 # the sound file still loads, and searches agree with a fresh build
        a = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT)[0]
        e.build_index()
        assert np.array_equal(a, e.search_knn(q, k, mode=vc.MODE_MIH_EXACT)[0])
