"""GPU against the committed fixtures only (no oracle call): tests/golden/search_fixtures.json."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixtures():
    with open(os.path.join(ROOT, "tests", "golden", "search_fixtures.json")) as f:
        return json.load(f)["fixtures"]


@pytest.mark.parametrize("idx", range(6))
def test_gpu_matches_committed_fixture(vc, idx):
    fx = _fixtures()[idx]
    k = fx["k"]
    q = np.stack([np.frombuffer(bytes.fromhex(h), dtype=np.uint8) for h in fx["queries"]])
    with vc.Engine(fx["bits"], capacity=fx["n"], n_tables=fx["m"]) as e:
        e.add_synthetic(fx["n"], fx["seed"], fx["kind"], fx["n_centres"], fx["max_flips"])
        got, cnt = e.search_knn(q, k, mode=vc.MODE_LINEAR)
        for i in range(len(q)):
            assert [int(v) for v in got[i, : cnt[i]]] == fx["linear"][i]
        if fx["mih_exact"]:
            e.build_index()
            got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_EXACT, with_stats=True)
            for i, ex in enumerate(fx["mih_exact"]):
                assert [int(v) for v in got[i, : cnt[i]]] == ex["result"]
                assert (st[i].radius, st[i].n_sub_reads, st[i].n_candidates) == (ex["radius"], ex["n_sub_reads"], ex["n_candidates"])
            got, cnt, st = e.search_knn(q, k, mode=vc.MODE_MIH_APPROX, with_stats=True)
            for i, ex in enumerate(fx["mih_approx"]):
                assert [int(v) for v in got[i, : cnt[i]]] == ex["result"]
                assert (st[i].radius, st[i].n_candidates) == (ex["radius"], ex["n_candidates"])
