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


def test_driver_file_queries_match_oracle(oracle, tmp_path):
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


def test_driver_query_by_id(oracle, tmp_path):
    """image_search_client::search_image_by_id: the path the reference ships dead (distributed_image_search.cc:116)."""
    n, bits, m, k = 20000, 128, 4, 5
    codes = oracle.gen_codes(n, bits, 7, kind=1, n_centres=100, max_flips=6)
    (tmp_path / "lsh.code").write_bytes(codes.tobytes())
    out = _run([tmp_path / "lsh.code", n, bits, bits // m, k, "pilaf", 0, 0, 1234])
    pairs = [(int(a), int(b)) for a, b in re.findall(r"^(\d+) : (\d+)$", out, flags=re.M)]
    exp = oracle.linear_knn(codes, codes[1234], k)
    assert sorted(d for _, d in pairs) == [int(x >> SH) for x in exp]
    assert pairs[-1][1] == 0 and (1234, 0) in pairs     # nearest last: the image itself at distance 0
