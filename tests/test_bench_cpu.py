"""bench.py's multi-rank plumbing rehearsed on CPU: world size 2 over gloo, the GPU backend replaced by the oracle-backed
test double of test_sharded_cpu.py (the double lives under tests/; bench.py itself imports oracle/ only inside its
cpu_baseline functions).  Checked: rank/world handling, shard ranges, the bucketed exchange + flush() inside the timed
region, exactly one JSON line on rank 0 naming BASELINE configs[3], and a non-zero exit when the result check fails."""
import io
import json
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from test_sharded_cpu import OracleBackend  # noqa: E402


class _Timing:
    scan_ms = 0.0
    scan_launches = 0
    scan_bytes = 0


class BenchDouble(OracleBackend):
    """the GpuBackend surface bench.py uses, on the CPU; `corrupt` makes every reported distance one too large"""

    def __init__(self, vo, bits, lo, corrupt=False):
        super().__init__(vo, bits, lo)
        self.corrupt = corrupt

    def local_topk(self, queries, k, out, counts, mode):
        super().local_topk(queries, k, out, counts, mode)
        if self.corrupt:
            out += 1 << 32

    def timing(self):
        return _Timing()

    def get_code(self, gid):
        return self.codes[gid - self.id_base]

    def unrecovered(self):
        return 0


def _worker(rank, world, port, corrupt, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ["VC_BENCH_BUCKET"] = "3"                      # 5 steps = one full bucket + a partial one finished by flush()
    import bench
    from oracle import vc_oracle as vo
    buf = io.StringIO()
    rc = bench.main(["--gpus", str(world), "--steps", "5", "--warmup", "1", "--db-size", "6001", "--k", "20", "--queries", "4",
                     "--cpu-seconds", "0"],
                    backend_factory=lambda bits, lo, hi: BenchDouble(vo, bits, lo, corrupt=corrupt and rank == 1),
                    device_kind="cpu", dist_backend="gloo", out=buf)
    ret[rank] = (rc, buf.getvalue())


def _run(corrupt):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, corrupt, ret), nprocs=2, join=True)
    return dict(ret)


def test_bench_two_ranks_emit_one_line_for_configs3(oracle):
    ret = _run(False)
    assert ret[0][0] == 0 and ret[1][0] == 0
    assert ret[1][1] == ""                                   # only rank 0 prints
    lines = [ln for ln in ret[0][1].splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 1 and j["scaling"] == "strong"
    assert "configs[3]" in j["config"]["workload"] and "sharded 2 ways" in j["config"]["workload"]
    assert j["config"]["exchange"] == "inline, 3 batches per all-gather"
    assert j["results_check"] == "ok" and j["value"] > 0 and j["unit"] == "queries/s"
    assert j["metric"].startswith("queries/sec (k-NN top-100)") and "roofline" in j and "cpu_baseline" not in j


def test_bench_exits_nonzero_when_the_result_check_fails(oracle):
    ret = _run(True)                                         # rank 1 reports distances that its stored codes contradict
    assert ret[0][0] != 0 and ret[1][0] != 0                 # every rank learns it (all-reduce of the check)
    j = json.loads(ret[0][1].strip())
    assert j["results_check"] == "FAILED"
